"""Three launches of the bf16x6 gated attention kernel (attn_x6_wide64p_kernel) and of its fp32 twin at an M-frame bank (N = 1674), for
rocprofv3 --pmc passes.  (profiles/r06_gated64_pmc.txt was taken while the 32-query kernel of rounds 3-5 still existed beside it.)
    python tools/dev/pmc_gated_x6.py [M]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, 'aot-benchmark_amd'))
import torch, aot_hip
aot_hip.load()
from networks.layers.attention import gated_splits, gated_splits_x6
M = int(sys.argv[1]) if len(sys.argv) > 1 else 14
N, E = 1674, 1024
q = torch.randn(N, 128, device='cuda'); k = torch.randn(M * N, 128, device='cuda'); v = torch.randn(M * N, E, device='cuda')
u = torch.randn(N, E, device='cuda'); out = torch.empty(N, E, device='cuda'); part = torch.empty(16 * N * (E + 8), device='cuda')
bank = aot_hip.x6_gated_bank(1, M * N, 128, E, 'cuda')
aot_hip.gated_pack_x6(k, v, bank, M * N)
T = M * N
n64, n32 = gated_splits_x6(N, 1, T), gated_splits(N, T)
for _ in range(3):
    aot_hip.gated_attention_x6(q, bank, u, out, T, 128 ** 0.5, part=part, nsplit=n64)
    aot_hip.gated_attention(q, k, v, u, out, T, 128 ** 0.5, part=part, nsplit=n32)
torch.cuda.synchronize()
print('M', M, 'splits', n64, n32)
