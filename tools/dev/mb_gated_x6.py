"""bf16x6 gated (DeAOT) attention (aot_gated_attn_x6_f32 on packed K / V banks) against an fp64 reference: error and launch time
(kernel + merge) at several bank sizes and grid-level key splits (N = 1674 queries, value 1024 wide), beside the fp32 kernel.
(profiles/r06_gated64_first.txt / r06_gated64_pipelined.txt were taken with the 32-query kernel of rounds 3-5 in the second column.)
    python tools/dev/mb_gated_x6.py [path/to/libaot_hip.so] [quick] [N]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, 'aot-benchmark_amd'))
import torch, aot_hip
if len(sys.argv) > 1 and sys.argv[1]:
    aot_hip.LIB_PATH = os.path.abspath(sys.argv[1])
aot_hip.load()
from networks.layers.attention import gated_splits, gated_splits_x6
QUICK = len(sys.argv) > 2 and sys.argv[2] == 'quick'
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1674
E, MMAX = 1024, 14
g = torch.Generator(device='cuda').manual_seed(1)
q = torch.randn(N, 128, device='cuda', generator=g)
k = torch.randn(MMAX * N, 128, device='cuda', generator=g); v = torch.randn(MMAX * N, E, device='cuda', generator=g)
u = torch.randn(N, E, device='cuda', generator=g)
out6, out5 = torch.empty(N, E, device='cuda'), torch.empty(N, E, device='cuda')
part = torch.empty(16 * N * (E + 8), device='cuda')
bank = aot_hip.x6_gated_bank(1, MMAX * N, 128, E, 'cuda')
for slot in range(MMAX):
    aot_hip.gated_pack_x6(k[slot * N:(slot + 1) * N], v[slot * N:(slot + 1) * N], bank, N, slot=slot)
torch.cuda.synchronize()


def timed(run, n=15):
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for M in ((1, 4, 14) if QUICK else (1, 2, 4, 8, 14)):
    T = M * N if M != 2 else 2 * N - 13
    ref = (torch.softmax((q.double() / 128 ** 0.5) @ k[:T].double().t(), -1) @ v[:T].double()) * u.double()
    tiles = (T + 31) // 32
    pick64, pick32 = gated_splits_x6(N, 1, T), gated_splits(N, T)
    for ns in sorted({1, 2, 4, 6, 8, 9, 12, 16, pick64, pick32}):
        if ns > max(1, tiles // 4):
            continue
        pt = part if ns > 1 else None
        new = lambda: aot_hip.gated_attention_x6(q, bank, u, out6, T, 128 ** 0.5, part=pt, nsplit=ns)
        old = lambda: aot_hip.gated_attention(q, k, v, u, out5, T, 128 ** 0.5, part=pt, nsplit=ns)
        t6, t5 = timed(new), timed(old)
        first = out6.clone(); new(); torch.cuda.synchronize()
        rep = float((first - out6).abs().max())
        e6, e5 = float((out6.double() - ref).abs().max()), float((out5.double() - ref).abs().max())
        gf = 2.0 * N * T * 1152
        print('M=%2d T=%5d ns=%2d%s%s q64 %7.1f us (%5.1f TF-eq, %.3f of 416.7)  fp32 %7.1f us (%5.1f TF)  x%.2f   max err vs fp64: q64 %.2e  fp32 %.2e  (q64 run-to-run %.1e)'
              % (M, T, ns, '*' if ns == pick64 else ' ', '+' if ns == pick32 else ' ', t6, gf / t6 * 1e-6, gf / t6 * 1e-6 / 416.7, t5, gf / t5 * 1e-6,
                 t5 / t6, e6, e5, rep), flush=True)
print('pack of one frame: K %.1f us, V %.1f us' % (timed(lambda: aot_hip.gated_pack_x6(k[:N], None, bank, N, slot=3)),
                                                      timed(lambda: aot_hip.gated_pack_x6(None, v[:N], bank, N, slot=3))))
