"""Times the gated (DeAOT) attention launch, N = 1674 queries, value 1024 wide, at several bank sizes and grid-level key splits:
    python tools/dev/mb_gated.py [path/to/libaot_hip.so]
Prints, per bank size M, the launch time (kernel + merge) for each split count and what gated_splits() picks."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, 'aot-benchmark_amd'))
import torch, aot_hip
if len(sys.argv) > 1 and sys.argv[1]:
    aot_hip.LIB_PATH = os.path.abspath(sys.argv[1])
aot_hip.load()
from networks.layers.attention import gated_splits
N, E = 1674, 1024
g = torch.Generator(device='cuda').manual_seed(1)
q = torch.randn(N, 128, device='cuda', generator=g); out = torch.empty(N, E, device='cuda')
k = torch.randn(14 * N, 128, device='cuda', generator=g); v = torch.randn(14 * N, E, device='cuda', generator=g)
u = torch.randn(N, E, device='cuda', generator=g)
part = torch.empty(16 * N * (E + 8), device='cuda')
for M in (1, 2, 4, 8, 14):
    T = M * N
    row = []
    for ns in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 16):
        if ns > max(1, (T // 32) // 4):
            continue
        run = lambda: aot_hip.gated_attention(q, k, v, u, out, T, 128 ** 0.5, part=part if ns > 1 else None, nsplit=ns)
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(15): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 15
        row.append('%d:%.0f' % (ns, us))
    pick = gated_splits(N, T)
    best = min(row, key=lambda r: float(r.split(':')[1]))
    print('M=%2d  GF %.1f  picked ns=%d  best %s us (%.0f TF) | %s' % (M, 2.0 * N * T * 1152 / 1e9, pick, best,
          2.0 * N * T * 1152 / float(best.split(':')[1]) * 1e-6, ' '.join(row)), flush=True)
