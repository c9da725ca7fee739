"""DeAOT's windowed gated propagation (aot_local_gated_f32: scores + softmax + aggregate) at 480p: time of the three launches together and
a checksum of the output, for A/B runs of library variants.     python tools/dev/mb_local_gated.py [path/to/libaot_hip.so] [h] [w]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, 'aot-benchmark_amd'))
import torch, torch.nn.functional as F, aot_hip
if len(sys.argv) > 1 and sys.argv[1]:
    aot_hip.LIB_PATH = os.path.abspath(sys.argv[1])
aot_hip.load()
h, w = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (31, 54)
E, N = 1024, h * w
g = torch.Generator(device='cuda').manual_seed(3)
r = lambda *s: torch.randn(*s, device='cuda', generator=g)
gq, gk, gv, gu, go = r(N, 128) * 1.5, r(N, 128) * 1.5, r(N, E), r(N, E), torch.empty(N, E, device='cuda')
relw, relb = r(225, 128) * 0.2, r(225) * 0.3
gtk = F.pad((relw.view(15, 15, 128) * 128 ** 0.5).permute(0, 2, 1), (0, 1)).contiguous()
gtb = F.pad(relb.view(15, 15), (0, 1)).contiguous()
prob = torch.empty(225 * N, device='cuda')
run = lambda: aot_hip.local_gated(gq, gk, gv, gu, gtk, gtb, prob, go, h, w, 128 ** 0.5)
for _ in range(5): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 50
e0.record()
for _ in range(n): run()
e1.record(); torch.cuda.synchronize()
print('%dx%d: %.1f us per call (scores + softmax + aggregate); checksum %.9e  max %.6e' % (h, w, e0.elapsed_time(e1) * 1e3 / n, float(go.double().sum()), float(go.abs().max())))
