import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, 'aot-benchmark_amd'))
import torch, aot_hip
aot_hip.load()
H, C = 8, 256
for Nq, T, ns in ((1674, 1674, 5), (1674, 5035, 3), (1674, 1674, 2), (500, 1674, 5)):
    g = torch.Generator().manual_seed(Nq + T)
    q = (torch.randn(Nq, C, generator=g) * 2).cuda(); k = (torch.randn(T, C, generator=g) * 2).cuda(); v = torch.randn(T, C, generator=g).cuda()
    bank = aot_hip.x6_bank(1, T + 40, C, 'cuda')
    aot_hip.attention_pack_x6(k, v, bank, T, slot=0)
    o6, o32 = torch.empty(Nq, C, device='cuda'), torch.empty(Nq, C, device='cuda')
    p6 = torch.zeros(ns * Nq * (C + 2 * H), device='cuda'); p32 = torch.zeros_like(p6)
    for rep in range(6):
        aot_hip.attention_x6(q, bank, o6, T, H, 32 ** 0.5, part=p6, nsplit=ns)
        aot_hip.attention(q, k, v, o32, T, H, 32 ** 0.5, part=p32, nsplit=ns)
        torch.cuda.synchronize()
        d = (o6 - o32).abs()
        O6, O32 = p6[:ns * Nq * C].view(ns, Nq, H, 32), p32[:ns * Nq * C].view(ns, Nq, H, 32)
        ml6, ml32 = p6[ns * Nq * C:].view(ns, Nq, H, 2), p32[ns * Nq * C:].view(ns, Nq, H, 2)
        dm = (ml6[..., 0] - ml32[..., 0]).abs(); dl = (ml6[..., 1] - ml32[..., 1]).abs() / ml32[..., 1].abs().clamp(min=1e-20)
        dO = (O6 - O32).abs().amax(-1)
        badq = (d.amax(1) > 1e-4).nonzero().flatten()
        print('Nq %d T %d ns %d rep %d: max|x6-fp32| %.2e; bad query rows %d (first %s); per split: max dm %s  max rel dl %s  max dO %s'
              % (Nq, T, ns, rep, float(d.max()), badq.numel(), badq[:6].tolist(), ['%.1e' % float(x) for x in dm.amax((1, 2))],
                 ['%.1e' % float(x) for x in dl.amax((1, 2))], ['%.1e' % float(x) for x in dO.amax((1, 2))]))
        if badq.numel():
            bad = (dO > 1e-3).nonzero()
            print('   bad (split, qrow, head) sample:', bad[:8].tolist(), ' #bad', bad.shape[0], ' heads:', sorted(set(bad[:, 2].tolist())),
                  ' splits:', sorted(set(bad[:, 0].tolist())), ' q tiles:', sorted(set((bad[:, 1] // 32).tolist()))[:20])
