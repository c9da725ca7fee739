"""The windowed (short-term) attention kernels of one 480p frame on the default stream, four launches each, for rocprofv3 --pmc passes
(VERDICT r3 next #8): local_attn_d32_kernel<7,8> (AOT, 31 x 54 tokens, 8 heads of 32) and lgp_{scores,softmax,aggregate}_kernel (DeAOT,
q / k 128 wide, v 1024 wide).
    rocprofv3 --kernel-trace --pmc <counters> -d out -o p -- python tools/dev/pmc_local.py"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, 'aot-benchmark_amd'))
import torch, torch.nn.functional as F, aot_hip
aot_hip.load()
h, w, H, C, E = 31, 54, 8, 256, 1024
N = h * w
g = torch.Generator(device='cuda').manual_seed(3)
r = lambda *s: torch.randn(*s, device='cuda', generator=g)
q, k, v, out = r(N, C) * 1.5, r(N, C) * 1.5, r(N, C), torch.empty(N, C, device='cuda')
tk, tb, tv = aot_hip.pack_local_tables((r(H * 225, 32, 1, 1) * 0.3).cpu(), (r(H * 225) * 0.3).cpu(), (r(H, 32, 225) * 0.2).cpu(), H)
tk, tb, tv = tk.cuda(), tb.cuda(), tv.cuda()
gq, gk, gv, gu, go = r(N, 128) * 1.5, r(N, 128) * 1.5, r(N, E), r(N, E), torch.empty(N, E, device='cuda')
relw, relb = r(225, 128) * 0.2, r(225) * 0.3
gtk = F.pad((relw.view(15, 15, 128) * 128 ** 0.5).permute(0, 2, 1), (0, 1)).contiguous()
gtb = F.pad(relb.view(15, 15), (0, 1)).contiguous()
prob = torch.empty(225 * N, device='cuda')
for _ in range(4):
    aot_hip.local_attention(q, k, v, tk, tb, tv, out, h, w, H, 32 ** 0.5)
    aot_hip.local_gated(gq, gk, gv, gu, gtk, gtb, prob, go, h, w, 128 ** 0.5)
torch.cuda.synchronize()
print('done: 4 launches of each windowed-attention kernel')
