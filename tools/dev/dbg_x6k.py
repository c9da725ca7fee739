"""debug: split-K on the 64x64 direct-weight kernel: slab contents against the partial products"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, 'aot-benchmark_amd'))
import torch, aot_hip
aot_hip.load()
torch.manual_seed(0)
for (M, K, N, ks) in ((256, 128, 64, 2), (300, 256, 128, 4), (1674, 1024, 256, 2)):
    x = torch.randn(M, K, device='cuda'); w = aot_hip.attach_wt(torch.randn(K, N, device='cuda') / K ** 0.5, K)
    out = torch.zeros(M, N, device='cuda'); ref = torch.zeros(M, N, device='cuda')
    aot_hip.X6_TILE = 66
    with aot_hip.use_gemm_table('throughput', 'bf16x6'):
        aot_hip.conv2d(x, w, None, ref, 1, M, K, 1, M, N)
    aot_hip.X6_TILE = 0
    aot_hip.conv2d_x6k(x, w, None, out, 1, M, K, 1, M, N, ksplit=-ks)
    torch.cuda.synchronize()
    slab = list(aot_hip._x6k_ws._bufs.values())[0][:ks * M * N].view(ks, M, N)
    nk = K // ks
    print('M %d K %d N %d ks %d: max |out - ref| %.3g ; x@w check %.3g' % (M, K, N, ks, float((out - ref).abs().max()), float((ref - x @ w).abs().max())))
    for s in range(ks):
        want = x[:, s * nk:(s + 1) * nk] @ w[s * nk:(s + 1) * nk]
        d = (slab[s] - want).abs()
        print('   slice %d: max err %.3g, rows with err > 1e-3: %d, cols: %d ; slab abs mean %.3g want %.3g' % (
            s, float(d.max()), int((d.max(1)[0] > 1e-3).sum()), int((d.max(0)[0] > 1e-3).sum()), float(slab[s].abs().mean()), float(want.abs().mean())))
    # which slice's data sits where?
    for s in range(ks):
        for s2 in range(ks):
            want = x[:, s2 * nk:(s2 + 1) * nk] @ w[s2 * nk:(s2 + 1) * nk]
            if float((slab[s] - want).abs().max()) < 1e-3: print('   slab %d holds the product of k-slice %d' % (s, s2))
