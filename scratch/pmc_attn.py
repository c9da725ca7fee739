"""attention M=14 launches for PMC collection"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'aot-benchmark_amd'))
import torch, aot_hip
aot_hip.load()
N, C, H = 1674, 256, 8
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 12
q = torch.randn(N, C, device='cuda')
T = 14 * N; k, v = torch.randn(T, C, device='cuda'), torch.randn(T, C, device='cuda'); out = torch.empty(N, C, device='cuda')
part = torch.empty(ns * N * (C + 2 * H), device='cuda')
for _ in range(3): aot_hip.attention(q, k, v, out, T, H, 32 ** 0.5, part=part, nsplit=ns)
torch.cuda.synchronize()
