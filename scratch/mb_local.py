import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'aot-benchmark_amd'))
import torch, aot_hip
if len(sys.argv) > 1: aot_hip.LIB_PATH = sys.argv[1]
aot_hip.load()
h, w, H, C = 31, 54, 8, 256
N = h * w
g = torch.Generator().manual_seed(0)
q, k, v = (torch.randn(N, C, generator=g).cuda() for _ in range(3))
tk, tb, tv = aot_hip.pack_local_tables(torch.randn(H * 225, 32, generator=g), torch.randn(H * 225, generator=g), torch.randn(H, 32, 225, generator=g), H)
tk, tb, tv = tk.cuda(), tb.cuda(), tv.cuda()
out = torch.empty(N, C, device='cuda')
def run(): aot_hip.local_attention(q, k, v, tk, tb, tv, out, h, w, H, 32 ** 0.5)
for _ in range(5): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 50
print('local_attn %dx%d: %.1f us  (%.1f TFLOP/s on 4*2*N*225*C = %.2f GF)' % (h, w, us, 4 * 2 * N * 225 * C / us / 1e6, 4 * 2 * N * 225 * C / 1e9))
