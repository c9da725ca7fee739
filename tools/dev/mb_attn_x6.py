"""bf16x6 attention (aot_attn_x6_f32 on a packed bank) against the fp32 kernel and an fp64 reference: error and launch time at
several bank sizes (N = 1674 queries, 8 heads, the bank appended frame by frame through aot_attn_pack_x6_f32).
    python tools/dev/mb_attn_x6.py [path/to/libaot_hip.so] [sweep]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, 'aot-benchmark_amd'))
import torch, aot_hip
if len(sys.argv) > 1 and sys.argv[1]:
    aot_hip.LIB_PATH = os.path.abspath(sys.argv[1])
aot_hip.load()
from networks.layers.attention import attn_splits
N, C, H, MMAX = 1674, 256, 8, 14
g = torch.Generator(device='cuda').manual_seed(1)
q = torch.randn(N, C, device='cuda', generator=g) * 2.0
k = torch.randn(MMAX * N, C, device='cuda', generator=g); v = torch.randn(MMAX * N, C, device='cuda', generator=g)
out, out6 = torch.empty(N, C, device='cuda'), torch.empty(N, C, device='cuda')
part = torch.empty(12 * N * (C + 2 * H), device='cuda')
bank = aot_hip.x6_bank(1, MMAX * N, C, 'cuda')
for slot in range(MMAX):                                     # frame by frame, as the engine appends
    aot_hip.attention_pack_x6(k[slot * N:(slot + 1) * N], v[slot * N:(slot + 1) * N], bank, N, slot=slot)
torch.cuda.synchronize()
SWEEP = len(sys.argv) > 2 and sys.argv[2] == 'sweep'
SWEEP2 = len(sys.argv) > 2 and sys.argv[2] == 'sweep2'       # round 5: wider grid splits of the x6 kernel (three waves per SIMD)
QUICK = len(sys.argv) > 2 and sys.argv[2] == 'quick'       # A/B of kernel variants: two bank sizes, planned split, no extras


def timed(run, n=40):
    for _ in range(5): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def ref64(T):
    qh = (q.double() / 32 ** 0.5).view(N, H, 32).permute(1, 0, 2)
    kh = k[:T].double().view(T, H, 32).permute(1, 2, 0)
    vh = v[:T].double().view(T, H, 32).permute(1, 0, 2)
    return (torch.softmax(qh @ kh, -1) @ vh).permute(1, 0, 2).reshape(N, C)


for M in ((4, 14) if QUICK else (2, 4, 7, 10, 14) if SWEEP2 else (1, 2, 4, 8, 14)):
    T = M * N if M != 2 else 2 * N - 13            # (one ragged length: partial last tile)
    ref = ref64(T)
    for ns in ((1, 2, 3, 4) if SWEEP else (1, 2, 3, 4, 5, 6, 7, 9, 12) if SWEEP2 else (attn_splits(N, H, T, wg_waves=4),)):
        if ns > max(1, (T // 32) // 16):
            continue
        pt = part if ns > 1 else None
        f32 = lambda: aot_hip.attention(q, k, v, out, T, H, 32 ** 0.5, part=pt, nsplit=ns)
        x6 = lambda: aot_hip.attention_x6(q, bank, out6, T, H, 32 ** 0.5, part=pt, nsplit=ns)
        t32, t6 = timed(f32), timed(x6)
        first = out6.clone(); x6(); torch.cuda.synchronize()
        rep = float((first - out6).abs().max())
        e32, e6 = float((out.double() - ref).abs().max()), float((out6.double() - ref).abs().max())
        print('M=%2d T=%5d ns=%d  fp32 %7.1f us (%5.1f TF)  x6 %7.1f us (%5.1f TF-eq)  x%.2f   max err vs fp64: fp32 %.2e  x6 %.2e  (x6 run-to-run %.1e)'
              % (M, T, ns, t32, 4.0 * N * T * C / t32 * 1e-6, t6, 4.0 * N * T * C / t6 * 1e-6, t32 / t6, e32, e6, rep), flush=True)
if QUICK or SWEEP2:
    sys.exit(0)
# pack cost, two lanes, device-side slot
us = timed(lambda: aot_hip.attention_pack_x6(k[:N], v[:N], bank, N, slot=3))
print('pack of one frame (%d x %d): %.1f us (appended at slot 3: chunks straddling the neighbours are merged)' % (N, C, us))
bank2 = aot_hip.x6_bank(2, 3 * N, C, 'cuda')
kk = torch.randn(2 * 3 * N, C, device='cuda', generator=g); vv = torch.randn(2 * 3 * N, C, device='cuda', generator=g)
slot_dev = torch.zeros(1, dtype=torch.int32, device='cuda')
for slot in range(3):
    slot_dev.fill_(slot)
    src_k = torch.cat([kk[slot * N:(slot + 1) * N], kk[3 * N + slot * N:3 * N + (slot + 1) * N]])
    src_v = torch.cat([vv[slot * N:(slot + 1) * N], vv[3 * N + slot * N:3 * N + (slot + 1) * N]])
    aot_hip.attention_pack_x6(src_k, src_v, bank2, N, B=2, src_brows=N, slot_dev=slot_dev)
q2 = torch.randn(2 * N, C, device='cuda', generator=g); o_a, o_b = torch.empty(2 * N, C, device='cuda'), torch.empty(2 * N, C, device='cuda')
T = 3 * N - 40
aot_hip.attention(q2, kk, vv, o_a, T, H, 32 ** 0.5, B=2, kv_brows=3 * N)
aot_hip.attention_x6(q2, bank2, o_b, T, H, 32 ** 0.5, B=2)
print('two lanes, device slot: max |x6 - fp32| %.2e' % float((o_a - o_b).abs().max()))
