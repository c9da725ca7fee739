"""Times the long-term attention launch (N = 1674 queries, 8 heads) at several bank sizes for one libaot_hip build:
    python tools/dev/mb_attn.py [path/to/libaot_hip.so]        (A/B runs of kernel variants: one process per build)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, 'aot-benchmark_amd'))
import torch, aot_hip
if len(sys.argv) > 1 and sys.argv[1]:
    aot_hip.LIB_PATH = os.path.abspath(sys.argv[1])
aot_hip.load()
from networks.layers.attention import attn_splits
N, C, H = 1674, 256, 8
g = torch.Generator(device='cuda').manual_seed(1)
q = torch.randn(N, C, device='cuda', generator=g); out = torch.empty(N, C, device='cuda')
k = torch.randn(14 * N, C, device='cuda', generator=g); v = torch.randn(14 * N, C, device='cuda', generator=g)
part = torch.empty(4 * N * (C + 2 * H), device='cuda')
SWEEP = len(sys.argv) > 2 and sys.argv[2] == 'sweep'       # every grid-level split 1..4 instead of the planned one
res = []
for M in (1, 2, 4, 8, 14):
    T = M * N
    for ns in ((1, 2, 3, 4) if SWEEP else (attn_splits(N, H, T, wg_waves=4),)):
        if ns > max(1, (T // 32) // 16):
            continue
        run = lambda: aot_hip.attention(q, k, v, out, T, H, 32 ** 0.5, part=part if ns > 1 else None, nsplit=ns)
        for _ in range(5): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 40
        res.append('M=%d ns=%d %.1f us (%.0f TF)' % (M, ns, us, 4.0 * N * T * C / us * 1e-6))
print(os.path.basename(aot_hip.LIB_PATH), ' | '.join(res), ' checksum %.6f' % out.double().sum().item())
