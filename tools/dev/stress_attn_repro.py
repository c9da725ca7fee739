"""Stress run behind tests/test_parity_gpu.py::test_attention_kernels_reproducible_under_load: every flash-attention kernel launched
many times on full-size grids (several shapes and key splits, other kernels running concurrently on a second stream), every result
compared bit for bit with the first.    python tools/dev/stress_attn_repro.py [repeats]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, 'aot-benchmark_amd'))
import torch, aot_hip
aot_hip.load()
REP = int(sys.argv[1]) if len(sys.argv) > 1 else 60
N, C, H = 1674, 256, 8
g = torch.Generator(device='cuda').manual_seed(5)
side = torch.cuda.Stream()
noise_a = torch.randn(4096, 4096, device='cuda', generator=g); noise_b = torch.randn(4096, 4096, device='cuda', generator=g)
total = bad = 0
for M in (1, 3, 7, 14):
    T = M * N - (13 if M > 1 else 0)
    q = torch.randn(N, C, device='cuda', generator=g) * 2
    k = torch.randn(M * N, C, device='cuda', generator=g); v = torch.randn(M * N, C, device='cuda', generator=g)
    bank = aot_hip.x6_bank(1, M * N, C, 'cuda'); aot_hip.attention_pack_x6(k, v, bank, M * N)
    qg = torch.randn(N, 128, device='cuda', generator=g); kg = torch.randn(M * N, 128, device='cuda', generator=g)
    vg = torch.randn(M * N, 1024, device='cuda', generator=g); u = torch.randn(N, 1024, device='cuda', generator=g)
    gbank = aot_hip.x6_gated_bank(1, M * N, 128, 1024, 'cuda'); aot_hip.gated_pack_x6(kg, vg, gbank, M * N)
    for ns in sorted({1, 2, 3, 5, min(9, max(1, (T // 32) // 4))}):
        if ns > max(1, (T // 32) // 16):
            continue
        part = torch.empty(ns * N * (C + 2 * H), device='cuda'); partg = torch.empty(ns * N * (1024 + 8), device='cuda')
        runs = {
            'fp32 d32': (lambda o: aot_hip.attention(q, k, v, o, T, H, 32 ** 0.5, part=part, nsplit=ns), (N, C)),
            'x6 d32': (lambda o: aot_hip.attention_x6(q, bank, o, T, H, 32 ** 0.5, part=part, nsplit=ns), (N, C)),
            'fp32 gated': (lambda o: aot_hip.gated_attention(qg, kg, vg, u, o, T, 128 ** 0.5, part=partg, nsplit=ns), (N, 1024)),
            'x6 gated': (lambda o: aot_hip.gated_attention_x6(qg, gbank, u, o, T, 128 ** 0.5, part=partg, nsplit=ns), (N, 1024)),
        }
        for name, (run, shape) in runs.items():
            first = torch.empty(shape, device='cuda'); run(first)
            nb = 0
            for r in range(REP):
                if r % 3 == 0:
                    with torch.cuda.stream(side):
                        noise_a @ noise_b                      # something else on the chip
                o = torch.empty(shape, device='cuda'); run(o)
                nb += int(not torch.equal(first, o))
            total += REP; bad += nb
            if nb:
                print('M=%d ns=%d %-10s: %d of %d repeats differ' % (M, ns, name, nb, REP), flush=True)
torch.cuda.synchronize()
print('launches compared: %d, differing: %d' % (total, bad))
