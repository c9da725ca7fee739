"""conv_out behind the FPN head's conv_4x block (aot_gn_conv1x1_f32: GroupNorm-apply + ReLU on the A loads of the 128 -> 11 convolution at the 4x
map): time and checksum, for A/B runs of library variants.     python tools/dev/mb_convout.py [path/to/libaot_hip.so]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, 'aot-benchmark_amd'))
import torch, aot_hip
if len(sys.argv) > 1 and sys.argv[1]:
    aot_hip.LIB_PATH = os.path.abspath(sys.argv[1])
aot_hip.load()
from networks.layers.workspace import Workspace
M, K, N = 121 * 213, 128, 11
g = torch.Generator(device='cuda').manual_seed(5)
x = torch.randn(M, K, device='cuda', generator=g) * 1.5
gamma, beta = torch.randn(K, device='cuda', generator=g), torch.randn(K, device='cuda', generator=g)
w = torch.zeros(K, 12, device='cuda'); w[:, :N] = torch.randn(K, N, device='cuda', generator=g) / K ** 0.5
bias = torch.randn(N, device='cuda', generator=g)
ws = Workspace()
stats = aot_hip.groupnorm_stats(x, 8, aot_hip.gn_buffers(ws, x.device, 1, 8, 32), nsplit=32)
out = torch.zeros(M, 12, device='cuda')
run = lambda: aot_hip.gn_conv1x1(x, stats, gamma, beta, w, bias, out, 8, N, gn_act=aot_hip.ACT_RELU)
for _ in range(5): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 100
e0.record()
for _ in range(n): run()
e1.record(); torch.cuda.synchronize()
print('gn_conv1x1 %d x %d -> %d: %.1f us; checksum %.9e' % (M, K, N, e0.elapsed_time(e1) * 1e3 / n, float(out.double().sum())))
