import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'aot-benchmark_amd'))
import torch, aot_hip
if len(sys.argv) > 2: aot_hip.LIB_PATH = sys.argv[2]
aot_hip.load()
from networks.layers.attention import attn_splits
N, C, H = 1674, 256, 8
g = torch.Generator().manual_seed(0)
q = torch.randn(N, C, generator=g).cuda()
splits = [int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 and sys.argv[1] else [0]
for M in (1, 2, 4, 8, 14):
    T = M * N
    k, v = torch.randn(T, C, generator=g).cuda(), torch.randn(T, C, generator=g).cuda()
    if os.environ.get('DATA') == 'zeros': q.zero_(); k.zero_(); v.zero_()
    out = torch.empty(N, C, device='cuda')
    part = torch.empty(32 * N * (C + 2 * H), device='cuda')
    row = []
    for ns in splits:
        n = ns if ns > 0 else attn_splits(N, H, T)
        def run(): aot_hip.attention(q, k, v, out, T, H, 32 ** 0.5, part=part if n > 1 else None, nsplit=n)
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        if os.environ.get('CLK'):
            part[-4:].zero_(); run(); torch.cuda.synchronize(); cb = part[-4:].view(torch.int64).tolist(); print('   clk ratio %.2f -> %.0f MHz' % (cb[0] / max(cb[1], 1), cb[0] / max(cb[1], 1) * 100))
        row.append('ns=%2d %7.1f us %5.1f TF' % (n, us, 4.0 * N * T * C / us / 1e6))
    print('M=%2d T=%6d %.2f GF | ' % (M, T, 4.0 * N * T * C / 1e9) + ' | '.join(row))
