"""Per-shape microbenchmark of the implicit-GEMM conv for every conv/linear of the R50-AOTL 480p frame.
usage: python tools/dev/mb_gemm.py [cfgs e.g. -1,0,1,2,x6] [lib] [only] [batch]      (x6 = the bf16x6 family by shape, x6d / x6s = one member,
x6zN / x6kN = split-K N on the phase-shifted 128x128 / the 64x64 direct-weight kernel; the members x6n / x6w / x6r / x6o / x6p of rounds 3-5
are gone from the library)"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, 'aot-benchmark_amd'))
import torch, aot_hip
if len(sys.argv) > 2 and sys.argv[2]: aot_hip.LIB_PATH = os.path.abspath(sys.argv[2])
ONLY = sys.argv[3].split(',') if len(sys.argv) > 3 and sys.argv[3] else None
aot_hip.load()
cfgs = [c if c.startswith('x6') else int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else '-1').split(',')]
BATCH = int(sys.argv[4]) if len(sys.argv) > 4 else 1
# (name, H, W, Cin, Cout, K, stride, count per frame)
S = [('stem7x7s2', 481, 849, 4, 64, 7, 2, 1),
     ('l1.c1 64>64', 121, 213, 64, 64, 1, 1, 1), ('l1.c1 256>64', 121, 213, 256, 64, 1, 1, 2),
     ('l1.c2 3x3 64', 121, 213, 64, 64, 3, 1, 3), ('l1.c3 64>256', 121, 213, 64, 256, 1, 1, 4),
     ('l2.c1 256>128@4x', 121, 213, 256, 128, 1, 1, 1), ('l2.c2 3x3s2 128', 121, 213, 128, 128, 3, 2, 1),
     ('l2.c1 512>128', 61, 107, 512, 128, 1, 1, 3), ('l2.c2 3x3 128', 61, 107, 128, 128, 3, 1, 3),
     ('l2.c3 128>512', 61, 107, 128, 512, 1, 1, 4), ('l2.ds 256>512 s2', 121, 213, 256, 512, 1, 2, 1),
     ('l3.c1 512>256@8x', 61, 107, 512, 256, 1, 1, 1), ('l3.c2 3x3s2 256', 61, 107, 256, 256, 3, 2, 1),
     ('l3.c1 1024>256', 31, 54, 1024, 256, 1, 1, 5 + 3), ('l3.c2 3x3 256', 31, 54, 256, 256, 3, 1, 5 + 1),
     ('l3.c3 256>1024', 31, 54, 256, 1024, 1, 1, 6 + 3), ('l3.ds 512>1024 s2', 61, 107, 512, 1024, 1, 2, 1),
     ('lstt 256>512', 31, 54, 256, 512, 1, 1, 3), ('lstt 256>256', 31, 54, 256, 256, 1, 1, 12),
     ('lstt 512>256', 31, 54, 512, 256, 1, 1, 3), ('lstt 1024>256', 31, 54, 1024, 256, 1, 1, 0),
     ('dec ad8 512>256', 61, 107, 512, 256, 1, 1, 1), ('dec c8 3x3 256>128', 61, 107, 256, 128, 3, 1, 1),
     ('dec ad4 256>128', 121, 213, 256, 128, 1, 1, 1), ('dec c4 3x3 128', 121, 213, 128, 128, 3, 1, 1),
     ('dec out 128>11', 121, 213, 128, 11, 1, 1, 1)]
if os.environ.get('AOT_MB_SHAPES') == 'swin':      # the linears of the Swin-B trunk at 480 x 848 (tokens 120x212 / 60x106 / 30x53 / 15x27; depths 2, 2, 18, 2)
    S = []
    for st, (hh, ww, C, depth) in enumerate(((120, 212, 128, 2), (60, 106, 256, 2), (30, 53, 512, 18), (15, 27, 1024, 2))):
        S += [('s%d qkv %d>%d' % (st, C, 3 * C), 1, hh * ww, C, 3 * C, 1, 1, depth), ('s%d proj %d>%d' % (st, C, C), 1, hh * ww, C, C, 1, 1, depth),
              ('s%d fc1 %d>%d' % (st, C, 4 * C), 1, hh * ww, C, 4 * C, 1, 1, depth), ('s%d fc2 %d>%d' % (st, 4 * C, C), 1, hh * ww, 4 * C, C, 1, 1, depth)]
        if st < 3:
            S.append(('s%d merge %d>%d' % (st, 4 * C, 2 * C), 1, hh * ww // 4, 4 * C, 2 * C, 1, 1, 1))
tot = {c: 0.0 for c in cfgs}; totf = 0.0
print('%-20s %7s %5s %5s %8s | ' % ('shape', 'M', 'K', 'N', 'GF') + ' | '.join('cfg%3s us    TF' % c for c in cfgs))
for (name, H, W, Cin, Cout, K, s, cnt) in S:
    if ONLY and not any(o in name for o in ONLY): continue
    p = K // 2
    OH, OW = (H + 2 * p - K) // s + 1, (W + 2 * p - K) // s + 1
    M, KK = BATCH * OH * OW, K * K * Cin
    x = torch.randn(BATCH * H * W, Cin, device='cuda'); ldb = (Cout + 3) // 4 * 4
    w = torch.randn(KK, ldb, device='cuda') / KK ** 0.5; b = torch.randn(Cout, device='cuda')
    if KK % 32 == 0 and Cin % 32 == 0: w = aot_hip.attach_wt(w, Cin)
    out = torch.empty(M, ldb, device='cuda')
    gf = 2.0 * M * KK * Cout / 1e9
    row = []
    ref_out = None
    for c in cfgs:
        if (str(c).startswith('x6z') or str(c).startswith('x6k')) and len(c) > 3:      # x6zN = the phase-shifted 128x128 kernel with split-K N; x6kN = the 64x64 direct-weight kernel with split-K N
            ks = int(c[3:]) * (-1 if c.startswith('x6k') else 1)
            if Cin % 32 or (KK // 32) % abs(ks): row.append('      -      -'); continue
            aot_hip.pack_bf16x6(w)
            def run():
                aot_hip.conv2d_x6k(x, w, b, out, H, W, Cin, OH, OW, Cout, K, K, s, p, 1, act=1, B=BATCH, ksplit=ks)
        elif str(c).startswith('x6'):      # x6 = tile by shape, x6d = the 64x64 direct-weight kernel forced, x6s = the register-staged 128x128 kernel forced
            aot_hip.X6_TILE = {'x6': 0, 'x6d': 66, 'x6s': 129}[c]
            def run():        # (layers that do not qualify fall back to the fp32 dispatch inside conv2d, as in the engine)
                with aot_hip.use_gemm_table('throughput', 'bf16x6'):
                    aot_hip.conv2d(x, w, b, out, H, W, Cin, OH, OW, Cout, K, K, s, p, 1, act=1, B=BATCH)
        elif (c == 3 and Cout > 32) or (c >= 10 and Cin % 32): row.append('      -      -'); continue
        else:
          def run():
            aot_hip.conv2d_cfg(x, w, b, out, H, W, Cin, OH, OW, Cout, K, K, s, p, 1, act=1, cfg=c, wt=getattr(w, '_aot_wt', None), B=BATCH)
        for _ in range(3): run()
        if ref_out is None: ref_out = out[:, :Cout].clone()
        else:
            err = (out[:, :Cout] - ref_out).abs().max().item()
            if not err < 1e-3: print('   !! cfg', c, name, 'differs from first cfg by', err)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        tot[c] += us * cnt
        row.append('%7.1f %6.1f' % (us, gf / us * 1e-3 * 1e3 / 1e3 * 1e3 if False else gf * 1e3 / us))
    totf += gf * cnt
    print('%-20s %7d %5d %5d %8.3f | ' % (name, M, KK, Cout, gf) + ' | '.join(row) + '   x%d' % cnt)
print('per-frame total GF %.1f ; ' % totf + ' ; '.join('cfg%s: %.0f us (%.1f TF)' % (c, tot[c], totf * 1e3 / tot[c]) for c in cfgs))
