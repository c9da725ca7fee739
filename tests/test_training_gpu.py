"""Training-side slice on the device: the two losses with their gradients against the REAL reference's
networks/layers/loss.py under autograd (tests/golden/training_losses.npz), the AdamW step and the gradient clip against
torch's own, the EMA update against the reference's rule.  Run on the MI355X box."""
import json
import os

import numpy as np
import pytest
import torch

from common import GOLD, LOSS_CASES, loss_case_inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def hip():
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    import aot_hip
    aot_hip.load()
    return aot_hip


@pytest.mark.parametrize('case', sorted(LOSS_CASES))
def test_losses_and_gradients_match_reference(hip, case):
    """CrossEntropyLoss (hard-example mining at the start / middle / end of its annealing, and the plain mean) and
    SoftJaccordLoss (all classes present / some missing) on three samples with 4 / 11 / 2 classes and ignore bands: loss per
    sample within 2e-6 relative, d(sum_i i * loss_i)/d(logits) within 1e-6 of the reference's autograd."""
    from networks.layers.loss import CrossEntropyLoss, SoftJaccordLoss
    g = np.load(os.path.join(GOLD, 'training_losses.npz'))
    c = LOSS_CASES[case]
    logits, labels = loss_case_inputs(case)
    logits = [l.cuda().requires_grad_(True) for l in logits]
    fn = CrossEntropyLoss(c['top_k'], c['mining_steps']) if c['kind'] == 'ce' else SoftJaccordLoss()
    loss = fn(logits, [l.cuda() for l in labels], c['step'])
    w = torch.arange(1, len(logits) + 1, dtype=torch.float32, device='cuda')
    (loss * w).sum().backward()
    ref = g[case + '.loss']
    assert np.abs(loss.detach().cpu().numpy() - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max()), (loss, ref)
    for i, l in enumerate(logits):
        rg = g['%s.grad%d' % (case, i)]
        got = l.grad.cpu().numpy()
        assert got.shape == rg.shape
        assert np.abs(got - rg).max() <= 1e-6 * max(1e-3, np.abs(rg).max()) + 1e-9, (case, i, np.abs(got - rg).max(), np.abs(rg).max())
        if c['kind'] == 'ce':       # exactly the pixels the reference selected carry gradient
            assert ((np.abs(got).sum(1) > 0) == (np.abs(rg).sum(1) > 0)).all()


def test_adamw_and_clip_match_torch(hip):
    """utils.optim.AdamW (aot_adamw_step_f32) against torch.optim.AdamW over six steps with changing learning rates, per-group
    weight decay and the clip_grad_norm_ factor folded into the step (trainer.py:116-118,501-503)."""
    from utils.optim import AdamW
    torch.manual_seed(3)
    shapes = [(257, 33), (1000,), (3, 5, 7, 2)]
    ref_p = [torch.nn.Parameter(torch.randn(s)) for s in shapes]
    our_p = [torch.nn.Parameter(p.detach().clone().cuda()) for p in ref_p]
    wds = [0.07, 0.0, 0.001]
    ref = torch.optim.AdamW([{'params': [p], 'weight_decay': w} for p, w in zip(ref_p, wds)], lr=2e-4, weight_decay=0.07)
    ours = AdamW([{'params': [p], 'weight_decay': w, 'name': 'p%d' % i} for i, (p, w) in enumerate(zip(our_p, wds))], lr=2e-4,
                 weight_decay=0.07)
    for step in range(6):
        lr = 2e-4 * (1 + step) / 3
        for grp in ref.param_groups + ours.param_groups:
            grp['lr'] = lr
        for p, q in zip(ref_p, our_p):
            p.grad = torch.randn(p.shape) * (10.0 if step % 2 else 0.1)
            q.grad = p.grad.cuda()
        total_ref = torch.nn.utils.clip_grad_norm_(ref_p, 5.0)
        total, scale = ours.clip_grad_norm(5.0)
        assert total == pytest.approx(float(total_ref), rel=1e-6)
        ref.step()
        ours.step(grad_scale=scale)
        for p, q in zip(ref_p, our_p):
            assert (p.detach() - q.detach().cpu()).abs().max().item() <= 2e-6 * max(1.0, p.detach().abs().max().item())


def test_ema_update_matches_reference(hip):
    """utils.ema.ExponentialMovingAverage.update (aot_ema_update_f32): the shadow of a 5-element vector after each of three
    updates, as the reference's utils/ema.py computes it (tests/golden/training.json)."""
    from utils.ema import ExponentialMovingAverage
    with open(os.path.join(GOLD, 'training.json')) as f:
        gold = json.load(f)
    p = [torch.nn.Parameter(torch.arange(5, dtype=torch.float32, device='cuda'))]
    ema = ExponentialMovingAverage(p, decay=0.999)
    for t, want in enumerate(gold['ema_shadows'], start=1):
        with torch.no_grad():
            p[0].add_(0.5 * t)
        ema.update(p)
        assert torch.allclose(ema.shadow_params[0].cpu(), torch.tensor(want), rtol=1e-6, atol=1e-6)
    ema.store(p)
    ema.copy_to(p)
    assert torch.equal(p[0].data, ema.shadow_params[0])
    ema.restore(p)
    assert not torch.equal(p[0].data, ema.shadow_params[0])
