"""bf16x6 gated (DeAOT) attention (aot_gated_attn_x6_f32 on packed K / V banks) against the fp32 kernel and an fp64 reference:
error and launch time at several bank sizes and grid-level key splits (N = 1674 queries, value 1024 wide).
    python tools/dev/mb_gated_x6.py [path/to/libaot_hip.so] [quick]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, 'aot-benchmark_amd'))
import torch, aot_hip
if len(sys.argv) > 1 and sys.argv[1]:
    aot_hip.LIB_PATH = os.path.abspath(sys.argv[1])
aot_hip.load()
from networks.layers.attention import gated_splits
QUICK = len(sys.argv) > 2 and sys.argv[2] == 'quick'
N, E, MMAX = 1674, 1024, 14
g = torch.Generator(device='cuda').manual_seed(1)
q = torch.randn(N, 128, device='cuda', generator=g)
k = torch.randn(MMAX * N, 128, device='cuda', generator=g); v = torch.randn(MMAX * N, E, device='cuda', generator=g)
u = torch.randn(N, E, device='cuda', generator=g)
out, out6 = torch.empty(N, E, device='cuda'), torch.empty(N, E, device='cuda')
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


for M in ((4, 14) if QUICK else (1, 2, 4, 8, 14)):
    T = M * N if M != 2 else 2 * N - 13
    ref = (torch.softmax((q.double() / 128 ** 0.5) @ k[:T].double().t(), -1) @ v[:T].double()) * u.double()
    pick = gated_splits(N, T)
    for ns in sorted({1, pick, min(16, max(1, (T // 32) // 4)), 4} if not QUICK else {pick, 4, 5}):
        if ns > max(1, (T // 32) // 4):
            continue
        pt = part if ns > 1 else None
        f32 = lambda: aot_hip.gated_attention(q, k, v, u, out, T, 128 ** 0.5, part=pt, nsplit=ns)
        x6 = lambda: aot_hip.gated_attention_x6(q, bank, u, out6, T, 128 ** 0.5, part=pt, nsplit=ns)
        t32, t6 = timed(f32), timed(x6)
        first = out6.clone(); x6(); torch.cuda.synchronize()
        rep = float((first - out6).abs().max())
        e32, e6 = float((out.double() - ref).abs().max()), float((out6.double() - ref).abs().max())
        gf = 2.0 * N * T * 1152
        print('M=%2d T=%5d ns=%2d%s fp32 %7.1f us (%5.1f TF)  x6 %7.1f us (%5.1f TF-eq)  x%.2f   max err vs fp64: fp32 %.2e  x6 %.2e  (x6 run-to-run %.1e)'
              % (M, T, ns, '*' if ns == pick else ' ', t32, gf / t32 * 1e-6, t6, gf / t6 * 1e-6, t32 / t6, e32, e6, rep), flush=True)
print('pack of one frame: K %.1f us, V %.1f us' % (timed(lambda: aot_hip.gated_pack_x6(k[:N], None, bank, N, slot=3)),
                                                      timed(lambda: aot_hip.gated_pack_x6(None, v[:N], bank, N, slot=3))))
