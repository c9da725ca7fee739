"""Pins the CPU oracle against outputs of the REAL reference (tests/golden/*.npz, made by
tests/golden/make_golden.py from /root/reference).  Runs on CPU."""
import numpy as np
import pytest
import torch

from common import (GOLD, LOGIT_TOL, MHA_KNOB_CASES, case_clip, check_masks, load_case, mha_knob_inputs, run_teacher_forced,
                    synth_model_state)
from oracle.aot_oracle import OracleEngine, OracleInferEngine, OracleModel, mha_core


@pytest.mark.parametrize('case', ['c1_aott', 'c1b_aott_ragged', 'c1c_aotb', 'c2_r50_aotl', 'c2b_swinb_aotl', 'c2c_r101_aotl', 'c3a_deaott', 'c3b_r50_deaotl', 'c3c_swinb_deaotl', 'c3d_deaots'])
def test_oracle_matches_reference_golden(case):
    c, g = load_case(case)
    _, _, sd = synth_model_state(c['model'])
    frames, mask, objs, out_size = case_clip(c)
    eng = OracleEngine(OracleModel(c['model'], sd))
    keep = set(c['keep_logits'])
    res = run_teacher_forced(eng, frames, mask, objs, out_size, g, keep)
    no = c['num_obj'] + 1
    for t, (l4, m) in res.items():
        check_masks(m, g, t, 'oracle')
        if l4 is not None:
            err = np.abs(l4[:no] - g['logits4_%d' % t]).max()
            assert err < 1e-4, 'frame %d logits4 err %g' % (t, err)        # far inside the 1e-3 bar
            assert err < LOGIT_TOL


@pytest.mark.parametrize('case', ['c2_r50_aotl_70', 'c3_swinb_deaotl_480', 'c3b_r50_deaotl_70', 'c3_swinb_deaotl_480_70'])
def test_oracle_matches_reference_full_size(case):
    """BASELINE configs 2 and 3 at their full size against the REAL reference: the 70-frame R50-AOTL clip (bank M 1 -> 14;
    every mask, logits and last-layer LSTT output at frames 1 / 35 / 69) and SwinB-DeAOTL at 480x848 with 10 objects."""
    import os
    c, g = load_case(case)
    _, _, sd = synth_model_state(c['model'])
    frames, mask, objs, out_size = case_clip(c, g=g)
    # The R50-AOTL clip (what cpu_baseline and smoke() lean on) is replayed in full: all 69 frames, bank M 1 -> 14, logits +
    # LSTT output at frames 1 / 35 / 69 (~100 s of CPU).  The HIP path is checked against all 69 reference frames of every
    # whole-clip golden on the GPU (test_parity_gpu.py).
    # The two DeAOT 70-frame clips (round 3) are replayed over their first 11 frames (M 1 -> 3) by default; the record of a
    # full replay of all three 70-frame clips is committed as profiles/r03_oracle_full_clip.txt.
    cut = 70 if case == 'c2_r50_aotl_70' else 11
    full = os.environ.get('AOT_ORACLE_FULL_CLIP') == '1' or c['frames'] <= cut
    if not full:
        frames = frames[:cut]
    eng = OracleEngine(OracleModel(c['model'], sd))
    extra = {}
    res = run_teacher_forced(eng, frames, mask, objs, out_size, g, set(c['keep_logits']), extra=extra)
    assert len(res) == len(frames) - 1
    no = c['num_obj'] + 1
    for t, (l4, m) in res.items():
        check_masks(m, g, t, 'oracle')
        if l4 is not None:
            assert np.abs(l4[:no] - g['logits4_%d' % t]).max() < 1e-4
            ref = g['lstt_last_%d' % t]
            assert np.abs(extra['lstt_last_%d' % t] - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize('case', ['c4_aott_13obj', 'c4_r50_aotl_44obj', 'c4_deaott_44obj'])
def test_oracle_multi_group_matches_reference(case):
    """More than 10 objects (aot_engine.py:485-635): OracleInferEngine against the REAL reference's AOTInferEngine -- 13
    synthetic objects (2 groups) and the 44 / 43-object first-frame masks of datasets/Demo (5 groups): merged
    output-size logits (subsampled), the first group's stride-4 logits and every mask."""
    c, g = load_case(case)
    _, _, sd = synth_model_state(c['model'])
    frames, mask, objs, out_size = case_clip(c, g=g)
    assert int(mask.max()) == c['num_obj'] > 10
    eng = OracleInferEngine(OracleModel(c['model'], sd), long_term_mem_gap=c.get('gap'))
    extra = {}
    res = run_teacher_forced(eng, frames, mask, objs, out_size, g, set(c['keep_logits']), extra=extra, sub=c['sub'])
    assert len(eng.aot_engines) == -(-c['num_obj'] // 10)
    for t, (l4, m) in res.items():
        check_masks(m, g, t, 'oracle')
        if l4 is not None:
            assert np.abs(l4[:11] - g['logits4_%d' % t]).max() < 1e-4
            ref = g['merged_%d' % t]
            got = extra['merged_%d' % t]
            assert got.shape == ref.shape == (1 + 10 * len(eng.aot_engines),) + ref.shape[1:]
            # logit() of a clamped probability: the slope is 1/p(1-p) <= 1e5 at the clamp, so compare probabilities too
            mid = (ref > -6.9) & (ref < 6.9)                # 1e-3 < p < 1 - 1e-3: away from the clamp the logits themselves agree
            assert np.abs(got - ref)[mid].max() < 2e-3 and np.abs(got - ref).max() < 5e-2
            assert np.abs(1 / (1 + np.exp(-got)) - 1 / (1 + np.exp(-ref))).max() < 1e-5


def test_oracle_fp64_agrees_with_fp32():
    """The fp32 oracle's own rounding noise (vs an fp64 evaluation of the same weights) is ~1e-5: the
    1e-3 tolerance is a property of the algorithm, not of lucky summation order."""
    c, g = load_case('c1_aott')
    _, _, sd = synth_model_state(c['model'])
    frames, mask, objs, out_size = case_clip(c)
    outs = []
    for dt in (torch.float32, torch.float64):
        eng = OracleEngine(OracleModel(c['model'], sd, dtype=dt))
        with torch.no_grad():
            eng.add_reference_frame(frames[0].to(dt), mask.to(dt), objs)
            eng.match_propogate_one_frame(frames[1].to(dt))
            eng.decode_current_logits(out_size)
        outs.append(eng.pred_id_logits[0, :2].double())
    assert (outs[0] - outs[1]).abs().max().item() < 2e-4


@pytest.mark.parametrize('case', sorted(MHA_KNOB_CASES))
def test_oracle_attention_knobs_match_reference_module(case):
    """top_k / max_mem_len_ratio (attention.py:84-89,102-105): the oracle's mha_core against the REAL reference
    MultiheadAttention run with those constructor knobs (tests/golden/mha_knobs.npz)."""
    import os
    g = np.load(os.path.join(GOLD, 'mha_knobs.npz'))
    Q, K, V, H = mha_knob_inputs()
    sums = np.array([Q.double().sum().item(), K.double().sum().item(), V.double().sum().item()])
    assert np.allclose(sums, g['input_sums'], rtol=0, atol=1e-9), 'seeded inputs differ from the ones the golden was made with'
    out = mha_core(Q, K, V, H, **MHA_KNOB_CASES[case]).numpy()
    assert np.abs(out - g[case]).max() < 2e-6


def test_oracle_bounded_bank_keeps_first_and_most_recent():
    """long_term_mem_max (repo extension): the bank never exceeds the bound, always holds the reference frame's rows
    first, and an unbounded engine agrees with it until the bound is reached."""
    from utils.synth import synth_clip
    _, _, sd = synth_model_state('aott')
    frames, mask, objs, out_size = synth_clip(6, 6, (65, 81), (64, 80), 2)
    a = OracleEngine(OracleModel('aott', sd), long_term_mem_gap=1, long_term_mem_max=3)
    b = OracleEngine(OracleModel('aott', sd), long_term_mem_gap=1)
    with torch.no_grad():
        for e in (a, b):
            e.add_reference_frame(frames[0], mask, objs)
        first = a.long_term_memories[0][0].clone()
        for t in range(1, 6):
            outs = []
            for e in (a, b):
                e.match_propogate_one_frame(frames[t])
                outs.append(e.decode_current_logits(out_size))
                e.update_memory(mask)
            n = a.enc_hw
            assert a.long_term_memories[0][0].shape[0] == min(t + 1, 3) * n
            assert torch.equal(a.long_term_memories[0][0][:n], first)
            if t <= 3:      # frames 1..3 still see identical banks (bank of frame t is built from frames < t)
                assert torch.equal(outs[0], outs[1])
            else:
                assert not torch.equal(outs[0], outs[1])
            assert torch.equal(a.long_term_memories[0][0][-n:], b.long_term_memories[0][0][-n:])


def test_transforms_match_reference_classes():
    """MultiRestrictSize's size rule and sample order, MultiToTensor's normalisation (video_transforms.py:594-715):
    oracle restatement AND the product's host helper against the real reference classes (tests/golden/transforms.json)."""
    import json
    import os
    from oracle.aot_oracle import restrict_size as o_rs, to_tensor_normalise
    from utils.image import restrict_size as p_rs
    g = json.load(open(os.path.join(GOLD, 'transforms.json')))
    assert len(g['sizes']) >= 70
    for c in g['sizes']:
        kw = c['kw']
        scales = kw.get('multi_scale', [1.3])            # the class default (video_transforms.py:598)
        want = []
        for sc in scales:
            for rs in (o_rs, p_rs):
                hw = rs(c['h'], c['w'], kw.get('max_short_edge'), kw.get('max_long_edge', 800), sc, c['align_corners'])
                assert list(hw) == c['out'][len(want)][:2], (c, sc, hw)
            want.append([hw[0], hw[1], False])
            if kw.get('flip'):
                want.append([hw[0], hw[1], True])
        assert want == c['out']
    out = to_tensor_normalise(np.array(g['to_tensor_in'], np.float32))
    assert str(out.dtype) == g['to_tensor_dtype']
    assert np.array_equal(out.numpy(), np.array(g['to_tensor_out'], np.float32))     # bit-exact


def test_cubic_resize_restatement_vs_torch_bicubic():
    """cv2 is absent (parity unpinned for the cubic filter): the restated OpenCV algorithm is cross-checked against torch's
    independent bicubic, which implements the same a = -0.75 kernel and half-pixel mapping but computes the source
    coordinate in float32 (so noise images differ by ~gradient x 1e-4; smooth ones agree to float rounding)."""
    import torch.nn.functional as F
    from oracle.aot_oracle import cv2_cubic_resize
    rs = np.random.RandomState(0)
    yy, xx = np.meshgrid(np.arange(120, dtype=np.float32), np.arange(213, dtype=np.float32), indexing='ij')
    smooth = np.stack([128 + 100 * np.sin(yy / 17) * np.cos(xx / 23), 0.5 * yy + 0.3 * xx, 255 - 0.7 * xx], -1).astype(np.float32)
    for img, tol in ((smooth, 2e-3), ((rs.rand(120, 213, 3) * 255).astype(np.float32), 5e-2)):
        for (oh, ow) in ((121, 209), (65, 97), (240, 431)):
            a = cv2_cubic_resize(img, oh, ow)
            b = F.interpolate(torch.from_numpy(img).permute(2, 0, 1)[None], size=(oh, ow), mode='bicubic',
                              align_corners=False)[0].permute(1, 2, 0).numpy()
            assert np.abs(a - b).max() < tol
    assert np.array_equal(cv2_cubic_resize(smooth, 120, 213), smooth)


def test_oracle_swin_ragged_input_matches_reference():
    """Swin-B trunk on a 98x131 input (sides not multiples of the 4x4 patch): the REAL reference's zero padding in
    PatchEmbed (swin_transformer.py:501-509), odd token grids in PatchMerging and padded windows at every stage --
    tests/golden/swin_ragged.npz (make_golden.make_swin_ragged)."""
    import os
    from oracle.aot_oracle import swin_features
    g = np.load(os.path.join(GOLD, 'swin_ragged.npz'))
    _, _, sd = synth_model_state('swinb_aotl')
    with torch.no_grad():
        feats = swin_features(sd, torch.from_numpy(g['x']))
    assert len(feats) in (3, 4)
    for i, f in enumerate(feats):
        assert tuple(f.shape) == tuple(g['shape_%d' % i])
        ref = g['feat_%d' % i]
        assert np.abs(f[0, ::3].numpy() - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize('case', ['topk40', 'topk1', 'ratio3', 'ratio3_topk300', 'dense'])
def test_oracle_gated_propagation_knobs_match_reference(case):
    """DeAOT's long-video knobs (attention.py:674-679 max_mem_len_ratio, :689-693 top_k) against the REAL reference module
    GatedPropagation (one 128-wide head, value 1024): tests/golden/gp_knobs.npz (make_golden.make_gp_knobs)."""
    import os
    from common import GP_KNOB_CASES, gp_knob_inputs, gp_knob_state
    from networks.layers.attention import GatedPropagation
    from oracle.aot_oracle import gated_propagation
    g = np.load(os.path.join(GOLD, 'gp_knobs.npz'))
    Q, K, V, U, size_2d = gp_knob_inputs()
    assert np.allclose(g['input_sums'], [t.double().sum().item() for t in (Q, K, V, U)], rtol=1e-12)
    mod = GatedPropagation(d_qk=256, d_vu=512, num_head=1, use_linear=False, d_att=128)
    sd = {'gp.' + k: v for k, v in gp_knob_state(mod.state_dict()).items()}
    with torch.no_grad():
        out = gated_propagation(sd, 'gp', Q, K, V, U, size_2d, 1, False, 128, **GP_KNOB_CASES[case])
    ref = g[case]
    assert np.abs(out.numpy() - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize('case', ['tf_aott', 'tf_aott_prev', 'tf_aott_shuffle', 'tf_deaott_prob', 'tf_r50_deaotl', 'tf_swinb_deaotl'])
def test_oracle_training_forward_matches_reference(case):
    """aot_engine.py:33-108 of the real reference (train_forward.npz): ground-truth / prediction / probability feedback,
    second self-memorising frame, shuffled identities; losses per frame and sample, masks outside the reference's near-ties."""
    from common import TRAIN_CFG, TRAIN_FWD_CASES, model_cfg, train_batch
    from oracle.aot_oracle import train_forward
    c = TRAIN_FWD_CASES[case]
    g = np.load(GOLD + '/train_forward.npz')
    _, _, sd = synth_model_state(c['model'])
    frames, masks, objs, perms = train_batch(case)
    with torch.no_grad():
        loss, frame_loss, pred = train_forward(OracleModel(c['model'], sd), frames, masks, objs, c['step'], TRAIN_CFG,
                                               use_prev_pred=c.get('use_prev_pred', False),
                                               enable_prev_frame=c.get('enable_prev_frame', False),
                                               use_prev_prob=c.get('use_prev_prob', False), perms=perms,
                                               long_term_mem_gap=model_cfg(c['model']).TRAIN_LONG_TERM_MEM_GAP)
    ref = g[case + '.masks']
    ties = np.unpackbits(g[case + '.ties'])[:ref.size].reshape(ref.shape).astype(bool)
    bad = pred.numpy() != ref
    assert int((bad & ~ties).sum()) == 0, 'masks differ outside near-ties: %d' % int((bad & ~ties).sum())
    np.testing.assert_allclose(frame_loss.numpy(), g[case + '.frame_loss'], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(float(loss), float(g[case + '.loss']), rtol=1e-4)


@pytest.mark.parametrize('case', ['tf_aott', 'tf_deaott_prob', 'tf_r50_deaotl', 'tf_swinb_deaotl'])
def test_oracle_training_gradients_match_reference(case):
    """The BACKWARD of the training step: gradients of the loss of aot_engine.py:33-108 with respect to every trainable
    parameter, from the real reference's `loss.backward()` (tests/golden/train_grads.npz: 105 / 108 parameters -- L2 norm, sum,
    64 sampled entries each, a few small tensors in full), against autograd through the oracle's train_forward.  The
    encoder's frozen part (TRAIN_ENCODER_FREEZE_AT = 2) carries no gradient in either.  This pins the oracle's backward;
    the HIP training path is to be held to the same fixture."""
    from common import TRAIN_CFG, TRAIN_FWD_CASES, check_grads_against_golden, model_cfg, train_batch
    from oracle.aot_oracle import train_forward
    c = TRAIN_FWD_CASES[case]
    g = np.load(GOLD + '/train_grads.npz')
    names = [str(n) for n in g[case + '.names']]
    _, _, sd = synth_model_state(c['model'])
    model = OracleModel(c['model'], sd)
    for k in names:
        model.sd[k].requires_grad_(True)
    frames, masks, objs, perms = train_batch(case)
    loss, _, _ = train_forward(model, frames, masks, objs, c['step'], TRAIN_CFG, use_prev_pred=c.get('use_prev_pred', False),
                               enable_prev_frame=c.get('enable_prev_frame', False),
                               use_prev_prob=c.get('use_prev_prob', False), perms=perms,
                               long_term_mem_gap=model_cfg(c['model']).TRAIN_LONG_TERM_MEM_GAP)
    np.testing.assert_allclose(float(loss.detach()), float(g[case + '.loss']), rtol=1e-4)
    loss.backward()
    check_grads_against_golden(case, {k: t.grad for k, t in model.sd.items() if t.requires_grad}, g)
    # parameters the reference leaves without a gradient (frozen encoder stages) have none here either
    assert all(not t.requires_grad for k, t in model.sd.items() if k not in names)


@pytest.mark.parametrize('case', ['single', 'tta'])
def test_oracle_sequence_eval_matches_reference_evaluator(case):
    """The oracle's restatement of the evaluator loop against the REAL `Evaluator.evaluating` run on the same scenario with
    its own dataset / transform / collation / fusion / feedback / save code (evaluator_loop.npz; cv2.resize there is the
    oracle's restated cubic, so the filter itself stays unpinned): every saved mask outside the reference's near-ties,
    fused probabilities of the last frame."""
    from common import EVAL_LOOP_CASES, evaluator_scenario
    from oracle.aot_oracle import sequence_eval
    c = EVAL_LOOP_CASES[case]
    g = np.load(GOLD + '/evaluator_loop.npz')
    _, _, sd = synth_model_state('aott')
    frames, labels, nums = evaluator_scenario()
    res = sequence_eval(OracleModel('aott', sd), frames, labels, nums, flip=c['flip'], multiscale=c['ms'], long_term_mem_gap=2)
    ref = g[case + '.masks']
    assert len(res) == ref.shape[0] == 3
    ties = np.unpackbits(g[case + '.ties'])[:ref.size].reshape(ref.shape).astype(bool)
    for t, (lab, prob) in enumerate(res):
        bad = lab.numpy().astype(np.uint8) != ref[t]
        assert int((bad & ~ties[t]).sum()) == 0, 'frame %d: %d pixels differ outside near-ties' % (t + 1, int((bad & ~ties[t]).sum()))
        assert int(bad.sum()) <= max(4, int(ties[t].sum()))
    assert (ref[1][5:25, 100:140] == 3).all()                       # the object injected at frame 2 is in the saved mask
    assert g[case + '.obj_idx_len'].tolist() == [3, 4, 4]           # the dataset hands over the new object id from frame 2 on
    err = np.abs(res[-1][1].numpy() - g[case + '.prob_last']).max()
    assert err < 2e-5, err


def test_oracle_new_object_group_mid_clip_matches_reference():
    """Objects 10..13 injected at frame 2 open a second object group (c5_aott_newgroup.npz, the real AOTInferEngine under the
    evaluator's call sequence): the new group's engine starts with its own frame counter and an empty bank, the first one
    memorises the frame again.  Teacher-forced on the reference's masks: merged logits and masks of every frame."""
    from common import NEWGROUP_CASE, newgroup_clip, run_newgroup
    c = NEWGROUP_CASE
    g = np.load(GOLD + '/c5_aott_newgroup.npz')
    _, _, sd = synth_model_state(c['model'])
    frames, first, new_label = newgroup_clip()
    eng = OracleInferEngine(OracleModel(c['model'], sd), long_term_mem_gap=c['gap'])
    gold = lambda t, lg: torch.from_numpy(g['masks'][t - 1].astype(np.float32))[None, None]
    logits = run_newgroup(eng, frames, first, new_label, gold)
    assert [lg.shape[1] for lg in logits] == g['n_channels'].tolist() == [11, 11, 21, 21, 21]
    assert [e.frame_step for e in eng.aot_engines] == [c['frames'] - 1, c['frames'] - 1 - c['inject']]
    ref = g['masks']
    ties = np.unpackbits(g['ties'])[:ref.size].reshape(ref.shape).astype(bool)
    for t, lg in enumerate(logits, start=1):
        err = np.abs(lg[0, :, ::2, ::2].numpy() - g['merged_%d' % t]).max()
        assert err < 2e-4, 'frame %d merged logits err %g' % (t, err)
        lab = torch.argmax(lg, 1)[0]
        if t == c['inject']:
            lab = torch.where(new_label[0, 0] == 0, lab, new_label[0, 0].long())
        bad = lab.numpy().astype(np.uint8) != ref[t - 1]
        assert int((bad & ~ties[t - 1]).sum()) == 0


@pytest.mark.parametrize('case', ['aott', 'deaott'])
def test_oracle_memory_schedule_options_match_reference(case):
    """short_term_mem_skip (2 / 3) and skip_long_term_update on every third frame through the real reference's engines
    (memsched.npz): the oracle, teacher-forced on the reference's masks, gives the same logits and bank length."""
    from common import MEMSCHED_CASES, run_memsched
    from oracle.aot_oracle import _set_skip
    from utils.synth import synth_clip
    c = MEMSCHED_CASES[case]
    g = np.load(GOLD + '/memsched.npz')
    _, _, sd = synth_model_state(c['model'])
    eng = OracleInferEngine(OracleModel(c['model'], sd), long_term_mem_gap=c['gap'])
    frames, mask, objs, out_size = synth_clip(c['clip'], c['frames'], c['in_size'], c['out_size'], c['num_obj'])
    gold = lambda t, lg: torch.from_numpy(g[case + '.masks'][t - 1].astype(np.float32))[None, None]

    class Skip:                                     # the engines exist only after the reference frame: set the option then
        def __getattr__(self, name):
            return getattr(eng, name)

        def add_reference_frame(self, *a, **k):
            eng.add_reference_frame(*a, **k)
            _set_skip(eng, c['skip'])
    logits = run_memsched(Skip(), frames, mask, objs, out_size, gold)
    ref = g[case + '.masks']
    ties = np.unpackbits(g[case + '.ties'])[:ref.size].reshape(ref.shape).astype(bool)
    for t, lg in enumerate(logits, start=1):
        err = np.abs(lg[0, :c['num_obj'] + 1, ::2, ::2].numpy() - g['%s.logits_%d' % (case, t)]).max()
        assert err < 1e-4, 'frame %d logits err %g' % (t, err)
        bad = torch.argmax(lg, 1)[0].numpy().astype(np.uint8) != ref[t - 1]
        assert int((bad & ~ties[t - 1]).sum()) == 0
    e0 = eng.aot_engines[0]
    assert e0.long_term_memories[0][0].shape[0] // e0.enc_hw == int(g[case + '.bank_frames']) == 3


@pytest.mark.parametrize('case', ['c2_r50_aotl_70', 'c3b_r50_deaotl_70', 'c3_swinb_deaotl_480_70'])
def test_fp64_reference_fixture_is_consistent(case):
    """tests/golden/<case>_fp64.npz (the REAL reference in double over the whole-clip golden, make_fp64_ties.py): one record per
    propagated frame; the pixels where the reference's fp32 and fp64 argmax disagree carry the OTHER id in fp64; and the noise level
    the GPU parity tests lean on is what the fixture says -- on the ResNet clips every such pixel lies inside the fp32 run's own 2e-4
    near-tie mask, on the Swin-B clip a quarter of them do not (the reference's fp32 run is that far from its fp64 run there)."""
    from common import load_case, unpack_gapmask
    from fp64_ties import classify_flips, load_fp64_ties
    c, g = load_case(case)
    f64 = load_fp64_ties(case)
    assert f64 is not None
    T = c['frames'] - 1
    assert f64['stats'].shape == (T, 5) and f64['free_diff'].shape == (T,)
    own, outside = 0, 0
    for t in range(1, T + 1):
        idx, gap, diff = f64['idx_%d' % t], f64['gap_%d' % t], f64['diff_%d' % t]
        assert len(idx) == len(gap) == len(f64['top1_%d' % t]) == len(f64['top2_%d' % t])
        assert (gap >= 0).all() and (gap < 2e-4).all() and (np.diff(idx) > 0).all()
        ref = g['masks'][t - 1].reshape(-1)
        tie = unpack_gapmask(g, t, g['masks'][t - 1].shape).reshape(-1)
        own += len(diff)
        outside += int((~tie[diff]).sum())
        # where the stored fp64 id is known (gap64 < 2e-4) it differs from the fp32 id on exactly those pixels
        pos = {int(i): n for n, i in enumerate(idx)}
        for px in diff:
            if int(px) in pos:
                assert f64['top1_%d' % t][pos[int(px)]] != ref[px]
        # a mask equal to the reference's has no flips; one that takes the fp64 id on the reference's own flips sides with fp64
        assert classify_flips(f64, t, ref, ref)['flips'] == 0
        if len(diff):
            alt = ref.copy()
            for px in diff:
                alt[px] = f64['top1_%d' % t][pos[int(px)]] if int(px) in pos else (ref[px] + 1) % 11
            cl = classify_flips(f64, t, alt, ref)
            assert cl['flips'] == cl['ref_undecided'] == cl['sides_with_fp64'] == len(diff)
    assert own == int(f64['stats'][:, 2].sum()) and outside == int(f64['stats'][:, 4].sum())
    if 'swinb' in case:
        assert own > 400 and outside > 100
    else:
        assert 20 <= own <= 60 and outside == 0


def test_demo_real_image_fixture_is_consistent():
    """tests/golden/demo_1001 (make_demo_e2e.py: the REAL reference's Evaluator on six 1080p JPEG frames of its own demo sequence, 44
    objects): the files decode to what the golden was made from, one mask + two near-tie maps per propagated frame, ids within 0..44,
    the tighter tie map inside the wider one.  (The GPU test runs SequenceEvaluator against it; the oracle's replay of this clip --
    five object groups at 577x1041 -- takes minutes on the host and is not part of the default CPU suite.)"""
    import os
    from PIL import Image
    from common import GOLD
    root = os.path.join(GOLD, 'demo_1001')
    g = np.load(os.path.join(root, 'golden.npz'))
    names = [str(n) for n in g['names']]
    assert len(names) == 6 and g['masks'].shape == (5, 1080, 1920) and str(g['model']) == 'r50_aotl'
    for n in names:
        im = Image.open(os.path.join(root, n))
        assert im.size == (1920, 1080) and im.mode == 'RGB'
    lab = np.array(Image.open(os.path.join(root, names[0].replace('jpg', 'png'))))
    assert lab.shape == (1080, 1920) and sorted(np.unique(lab).tolist()) == list(range(45))
    assert int(g['masks'].max()) <= 44 and tuple(g['input_size']) == (577, 1041) and tuple(g['logit_shape']) == (1, 51, 1080, 1920)
    n = 5 * 1080 * 1920
    wide, tight = np.unpackbits(g['ties_1e3'])[:n].astype(bool), np.unpackbits(g['ties_2e4'])[:n].astype(bool)
    assert not (tight & ~wide).any() and 0 < tight.sum() < wide.sum() < 0.02 * n
