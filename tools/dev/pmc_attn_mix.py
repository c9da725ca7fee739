"""The attention launches of one 70-frame R50-AOTL clip (bench.py's mix: per propagated frame and layer one self-attention
over the frame and one long-term attention over the bank, M = 1 + (t-1)//5 memorised frames), on the default stream, for
rocprofv3 --pmc passes (PMC collection hangs on bench.py's per-clip HIP streams)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, 'aot-benchmark_amd'))
import torch, aot_hip
aot_hip.load()
from networks.layers.attention import attn_splits
N, C, H = 1674, 256, 8
q = torch.randn(N, C, device='cuda'); out = torch.empty(N, C, device='cuda')
k = torch.randn(14 * N, C, device='cuda'); v = torch.randn(14 * N, C, device='cuda')
part = torch.empty(4 * N * (C + 2 * H), device='cuda')      # sized like MultiheadAttention.core's slab set
n = 0
for t in range(1, 70):
    M = 1 + (t - 1) // 5
    for layer in range(3):
        for T in (N, M * N):
            ns = attn_splits(N, H, T, wg_waves=4)
            aot_hip.attention(q, k, v, out, T, H, 32 ** 0.5, part=part if ns > 1 else None, nsplit=ns)
            n += 1
torch.cuda.synchronize()
print('launches', n)
