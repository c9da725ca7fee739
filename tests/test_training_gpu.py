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


@pytest.mark.parametrize('case', ['tf_aott', 'tf_aott_prev', 'tf_aott_shuffle', 'tf_deaott_prob', 'tf_r50_deaotl', 'tf_swinb_deaotl'])
def test_training_forward_matches_reference(hip, case):
    """AOTEngine.forward / DeAOTEngine.forward (aot_engine.py:33-108) against the REAL reference's training engine on the same
    seeded batch (tests/golden/train_forward.npz): ground-truth, prediction and probability feedback, the second
    self-memorising frame, shuffled identities.  Per-frame per-sample losses within 1e-4 relative (values 2..5; measured <= 2e-6), masks equal
    outside the reference's own near-tie pixels (a handful of knock-on flips allowed where predictions are fed back)."""
    from common import TRAIN_CFG, TRAIN_FWD_CASES, synth_model_state, train_batch
    from networks.engines import build_engine
    c = TRAIN_FWD_CASES[case]
    g = np.load(os.path.join(GOLD, 'train_forward.npz'))
    cfg, model, _ = synth_model_state(c['model'], cfg_overrides=TRAIN_CFG)
    model = model.cuda().eval()
    frames, masks, objs, perms = train_batch(case)
    engine = build_engine(cfg.MODEL_ENGINE, phase='train', aot_model=model, gpu_id=0,
                          long_term_mem_gap=cfg.TRAIN_LONG_TERM_MEM_GAP).eval()
    engine.restart_engine(len(objs), perms is not None)
    if perms is not None:
        assert all(sorted(p.tolist()) == list(range(11)) and int(p[0]) == 0 for p in engine.id_shuffle)   # what restart drew
        engine.id_shuffle = [p.cuda() for p in perms]
    with torch.no_grad():
        loss, pred, frame_loss, boards = engine(frames.cuda(), masks.cuda(), len(objs), objs, step=c['step'],
                                                use_prev_pred=c.get('use_prev_pred', False),
                                                enable_prev_frame=c.get('enable_prev_frame', False),
                                                use_prev_prob=c.get('use_prev_prob', False))
    ref = g[case + '.masks']
    assert len(pred) == len(frame_loss) == c['frames'] and boards == {'image': {}, 'scalar': {}}
    got = torch.stack(pred).cpu().numpy()
    assert got.shape == ref.shape and pred[0].dtype == torch.long
    ties = np.unpackbits(g[case + '.ties'])[:ref.size].reshape(ref.shape).astype(bool)
    bad = got != ref
    hard = int((bad & ~ties).sum())
    assert hard <= (8 if c.get('use_prev_pred') else 0), 'masks differ outside near-ties: %d' % hard
    fl = torch.stack(frame_loss).cpu().numpy()
    np.testing.assert_allclose(fl, g[case + ".frame_loss"], rtol=1e-4, atol=2e-5)       # measured on MI355X: <= 2e-6
    np.testing.assert_allclose(float(loss), float(g[case + '.loss']), rtol=1e-4)
    print('train_forward %s: loss %.6f (ref %.6f), max frame-loss err %.2e, mask flips %d (%d outside ties)'
          % (case, float(loss), float(g[case + '.loss']), np.abs(fl - g[case + '.frame_loss']).max(), int(bad.sum()), hard))


def test_probability_map_identity(hip):
    """MODEL_USE_PREV_PROB (aot_engine.py:309-313): the identity embedding of a soft map is the dense patch_wise_id_bank
    convolution -- against fp64 conv2d for both bank geometries (17x17/s16/p8 and 16x16/s16/p0); a one-hot map gives what
    the label-map gather gives; feeding assign_identity's result back as curr_id_emb equals feeding the map itself."""
    import torch.nn.functional as F
    from common import synth_model_state
    from networks.engines import build_engine
    for align, (H, W) in ((True, (129, 161)), (False, (128, 160))):
        cfg, model, sd = synth_model_state('aott', cfg_overrides=dict(MODEL_ALIGN_CORNERS=align))
        model = model.cuda().eval()
        gen = torch.Generator().manual_seed(11)
        prob = torch.softmax(torch.randn(1, 11, H, W, generator=gen) * 2, 1)
        conv = model.patch_wise_id_bank
        ref = F.conv2d(prob.double(), sd['patch_wise_id_bank.weight'].double(), sd['patch_wise_id_bank.bias'].double(),
                       stride=conv.stride, padding=conv.padding)
        h, w = ref.shape[-2:]
        got = model.id_emb_from_prob(prob.cuda(), (h, w))
        err = (got.cpu().double() - ref[0].flatten(1).t()).abs().max().item()
        assert err < 2e-5 * max(1.0, ref.abs().max().item()), (align, err)
    # engine level (AOTT): one-hot map == label map; curr_id_emb == the map it was made from
    cfg, model, sd = synth_model_state('aott')
    model = model.cuda().eval()
    from utils.synth import synth_clip
    frames, mask, objs, _ = synth_clip(3, 3, (129, 161), (129, 161), 3, device='cuda')
    outs = []
    for mode in ('label', 'onehot', 'id_emb', 'soft'):
        e = build_engine(cfg.MODEL_ENGINE, phase='eval', aot_model=model, gpu_id=0, long_term_mem_gap=1)
        e.restart_engine()
        e.add_reference_frame(frames[0], mask, objs, frame_step=0)
        e.match_propogate_one_frame(frames[1])
        lg = e.decode_current_logits((129, 161))
        lab = lg.argmax(1, keepdim=True).float()
        onehot = (lab == torch.arange(11, device='cuda').view(1, -1, 1, 1)).float()
        if mode == 'label':
            e.update_memory(lab)
        elif mode == 'onehot':
            e.update_memory(onehot)
        elif mode == 'soft':
            e.update_memory(torch.softmax(lg, 1))
        else:
            c0 = e._cohorts[0]
            c0.update_short_term_memory(None, curr_id_emb=c0.assign_identity(lab))
        e.match_propogate_one_frame(frames[2])
        outs.append(e.decode_current_logits((129, 161)).clone())
    assert (outs[0] - outs[1]).abs().max().item() < 2e-4          # gather vs dense convolution of the same one-hot map
    assert (outs[0] - outs[2]).abs().max().item() < 1e-5
    d = (outs[0] - outs[3]).abs().max().item()
    assert 1e-4 < d < 10.0, d                                       # the soft map is a different (finite) input


def test_prob_feedback_engine_vs_oracle(hip):
    """Probabilities fed back frame after frame (what the reference's evaluator sets out to do under MODEL_USE_PREV_PROB,
    evaluator.py:409-425) through SequenceEvaluator vs the oracle engine driven the same way."""
    import torch.nn.functional as F
    from common import synth_model_state
    from networks.managers.evaluator import SequenceEvaluator
    from oracle.aot_oracle import OracleEngine, OracleModel, cv2_cubic_resize, restrict_size, to_tensor_normalise
    cfg, model, sd = synth_model_state('deaott', cfg_overrides=dict(TEST_FLIP=False, TEST_MULTISCALE=[1], MODEL_USE_PREV_PROB=True,
                                                                   TEST_MAX_SHORT_EDGE=None, TEST_MAX_LONG_EDGE=800 * 1.3,
                                                                   TEST_LONG_TERM_MEM_GAP=2))
    model = model.cuda().eval()
    rs = np.random.RandomState(5)
    H, W = 97, 145                      # already a legal size: the evaluator's resize is the identity
    frames = [(rs.rand(H, W, 3) * 255).astype(np.float32) for _ in range(4)]
    lab0 = np.zeros((H, W), np.float32); lab0[20:60, 30:80] = 1; lab0[50:90, 90:140] = 2
    assert restrict_size(H, W, None, 800 * 1.3, 1.0, True) == (H, W)
    ev = SequenceEvaluator(cfg, model)
    got = ev.run([torch.from_numpy(f).cuda() for f in frames], {0: torch.from_numpy(lab0).cuda()}, {0: 2})
    eng = OracleEngine(OracleModel('deaott', sd), long_term_mem_gap=2)
    prep = lambda f: to_tensor_normalise(cv2_cubic_resize(f, H, W)).unsqueeze(0)
    with torch.no_grad():
        eng.add_reference_frame(prep(frames[0]), torch.from_numpy(lab0)[None, None], [2], frame_step=0)
        for t in (1, 2, 3):
            eng.match_propogate_one_frame(prep(frames[t]))
            lg = eng.decode_current_logits((H, W))
            prob = torch.softmax(lg, 1)
            top2 = torch.topk(lg[0], 2, 0)[0]
            sure = (top2[0] - top2[1]) > 1e-3
            agree = got[t - 1].cpu() == lg.argmax(1)[0].float()
            assert agree[sure].all(), 'frame %d: %d sure pixels differ' % (t, int((~agree[sure]).sum()))
            eng.update_memory(F.interpolate(prob, size=eng.input_size_2d, mode='nearest'))


def test_new_object_group_mid_clip_vs_reference_golden(hip):
    """(A parity test of the inference engine; it sits at the end of the GPU suite because it is the newest.)  Objects
    10..13 injected at frame 2 open a second object group: AOTInferEngine starts a second COHORT there (own frame counter,
    empty bank) while the first one memorises the frame again and switches to mask separation -- against the REAL reference's
    AOTInferEngine under the evaluator's call sequence (tests/golden/c5_aott_newgroup.npz), teacher-forced on its masks:
    merged logits within the 1e-3 bar, masks equal outside the reference's near-ties."""
    from common import LOGIT_TOL, NEWGROUP_CASE, newgroup_clip, run_newgroup, synth_model_state
    from networks.engines import build_engine
    c = NEWGROUP_CASE
    g = np.load(os.path.join(GOLD, 'c5_aott_newgroup.npz'))
    cfg, model, _ = synth_model_state(c['model'])
    model = model.cuda().eval()
    eng = build_engine(cfg.MODEL_ENGINE, phase='eval', aot_model=model, gpu_id=0, long_term_mem_gap=c['gap'])
    frames, first, new_label = newgroup_clip()
    gold = lambda t, lg: torch.from_numpy(g['masks'][t - 1].astype(np.float32))[None, None]
    logits = run_newgroup(eng, frames, first, new_label, gold, to_dev=lambda x: x.cuda())
    assert [lg.shape[1] for lg in logits] == g['n_channels'].tolist()
    assert len(eng._cohorts) == 2 and len(eng.aot_engines) == 2
    assert [co.frame_step for co in eng._cohorts] == [c['frames'] - 1, c['frames'] - 1 - c['inject']]
    ref = g['masks']
    ties = np.unpackbits(g['ties'])[:ref.size].reshape(ref.shape).astype(bool)
    worst = 0.0
    for t, lg in enumerate(logits, start=1):
        lg = lg.cpu()
        got, want = lg[0, :, ::2, ::2].numpy(), g['merged_%d' % t]
        live = want > -1e9                       # unused identities of a single group sit at -1e10 (+- an ulp of 1024)
        assert (got[~live] < -1e9).all()
        err = np.abs(got - want)[live].max()
        worst = max(worst, float(err))
        assert err < LOGIT_TOL, 'frame %d merged logits err %g' % (t, err)
        lab = torch.argmax(lg, 1)[0]
        if t == c['inject']:
            lab = torch.where(new_label[0, 0] == 0, lab, new_label[0, 0].long())
        bad = lab.numpy().astype(np.uint8) != ref[t - 1]
        assert int((bad & ~ties[t - 1]).sum()) == 0, 'frame %d: %d pixels differ outside near-ties' % (t, int((bad & ~ties[t - 1]).sum()))
    print('newgroup: max merged-logit err %.2e' % worst)
