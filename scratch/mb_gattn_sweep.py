"""ns sweep of the DeAOT gated (wide-value) attention. usage: mb_gattn_sweep.py [lib] [Ms]"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'aot-benchmark_amd'))
import torch, aot_hip
if len(sys.argv) > 1 and sys.argv[1]: aot_hip.LIB_PATH = os.path.abspath(sys.argv[1])
aot_hip.load()
from networks.layers.attention import attn_splits
Ms = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [1, 2, 4, 8, 14]
N, DQ, E = 1674, 128, 1024
q = torch.randn(N, DQ, device='cuda'); out = torch.empty(N, E, device='cuda'); gate = torch.randn(N, E, device='cuda')
part = torch.empty(16 * N * (E + 2 * 4), device='cuda')
for M in Ms:
    T = M * N
    k = torch.randn(T, DQ, device='cuda'); v = torch.randn(T, E, device='cuda')
    row = []
    for ns in range(1, 17):
        def run(): aot_hip.gated_attention(q, k, v, gate, out, T, DQ ** 0.5, part=part if ns > 1 else None, nsplit=ns)
        for _ in range(2): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 5
        e0.record()
        for _ in range(n): run()
        e1.record(); torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) * 1e3 / n)
    best = min(range(16), key=lambda i: row[i]); h = attn_splits(N, E // 256, T, occ=1, c0=1.0)
    fl = 2.0 * N * T * (DQ + E)
    print('M=%2d best ns=%2d %6.1f us %5.1f TF (heuristic ns=%2d %6.1f us) | ' % (M, best + 1, row[best], fl / row[best] / 1e6, h, row[h - 1])
          + ' '.join('%6.0f' % t for t in row), flush=True)
