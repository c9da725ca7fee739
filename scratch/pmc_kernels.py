"""A few representative launches for PMC collection (rocprofv3 --pmc ...)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'aot-benchmark_amd'))
import torch, aot_hip
aot_hip.load()
from networks.layers.attention import attn_splits
g = torch.Generator().manual_seed(0)
def conv(H, W, Cin, Cout, K, s, cfg=-1):
    p = K // 2; OH, OW = (H + 2 * p - K) // s + 1, (W + 2 * p - K) // s + 1
    x = torch.randn(H * W, Cin, device='cuda'); w = torch.randn(K * K * Cin, Cout, device='cuda') * 0.01; out = torch.empty(OH * OW, Cout, device='cuda')
    for _ in range(3): aot_hip.conv2d_cfg(x, w, None, out, H, W, Cin, OH, OW, Cout, K, K, s, p, 1, cfg=cfg)
conv(121, 213, 128, 128, 3, 1)        # dec c4: LDS 64x64 BK32, best case
conv(121, 213, 64, 64, 3, 1)          # l1.c2
conv(31, 54, 256, 256, 3, 1)          # l3.c2: direct KS=8
conv(31, 54, 1024, 256, 1, 1)         # l3.c1: direct KS=4/8
conv(61, 107, 128, 128, 3, 1)         # l2.c2
N, C, H = 1674, 256, 8
q = torch.randn(N, C, device='cuda')
for M in (1, 8):
    T = M * N; k, v = torch.randn(T, C, device='cuda'), torch.randn(T, C, device='cuda'); out = torch.empty(N, C, device='cuda')
    ns = attn_splits(N, H, T); part = torch.empty(ns * N * (C + 2 * H), device='cuda')
    for _ in range(3): aot_hip.attention(q, k, v, out, T, H, 32 ** 0.5, part=part, nsplit=ns)
torch.cuda.synchronize()
