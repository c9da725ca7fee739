"""ns sweep of the long-term attention: time for every (M, nsplit). usage: mb_attn_sweep.py [lib] [Ms]"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'aot-benchmark_amd'))
import torch, aot_hip
if len(sys.argv) > 1 and sys.argv[1]: aot_hip.LIB_PATH = os.path.abspath(sys.argv[1])
aot_hip.load()
Ms = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else list(range(1, 15))
N, H, C = 1674, 8, 256
q = torch.randn(N, C, device='cuda'); out = torch.empty(N, C, device='cuda')
part = torch.empty(16 * N * (C + 2 * H), device='cuda')
for M in Ms:
    T = M * N
    k = torch.randn(T, C, device='cuda'); v = torch.randn(T, C, device='cuda')
    row = []
    for ns in range(1, 17):
        def run(): aot_hip.attention(q, k, v, out, T, H, 32 ** 0.5, part=part if ns > 1 else None, nsplit=ns)
        for _ in range(2): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        e0.record()
        for _ in range(n): run()
        e1.record(); torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) * 1e3 / n)
    best = min(range(16), key=lambda i: row[i])
    print('M=%2d best ns=%2d %6.1f us %5.1f TF | ' % (M, best + 1, row[best], 4.0 * N * T * C / row[best] / 1e6) + ' '.join('%6.1f' % t for t in row), flush=True)
