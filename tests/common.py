"""Shared helpers for the parity tests."""
import importlib
import json
import os

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')

LOGIT_TOL = 1e-3          # BASELINE.json north_star: pre-softmax logits within 1e-3 of the reference
TIE_GAP = 2e-4            # reference pixels whose top-2 logit gap is below this are argmax near-ties


def cases():
    with open(os.path.join(GOLD, 'cases.json')) as f:
        return json.load(f)


def load_case(name):
    c = cases()[name]
    g = np.load(os.path.join(GOLD, name + '.npz'))
    return c, g


def model_cfg(name):
    return importlib.import_module('configs.models.' + name).ModelConfig()


def synth_model_state(name, cfg_overrides=None):
    """Keyed synthetic weights for model `name`, built from THIS package's parameter tree (whose keys and
    shapes are tested against the reference's in test_state_dict_layout)."""
    from networks.models import build_vos_model
    from utils.synth import synth_state_dict
    cfg = model_cfg(name)
    for k, v in (cfg_overrides or {}).items():
        setattr(cfg, k, v)
    model = build_vos_model(cfg.MODEL_VOS, cfg)
    sd = synth_state_dict(model.state_dict())
    model.load_state_dict(sd)
    return cfg, model, sd


def case_clip(c, device='cpu', g=None):
    """Synthetic clip of a golden case; cases made from a real label map (datasets/Demo) carry it as `first_mask`."""
    from utils.synth import synth_clip
    frames, mask, objs, out_size = synth_clip(c['clip'], c['frames'], tuple(c['in_size']), tuple(c['out_size']),
                                              c['num_obj'], device=device)
    if g is not None and 'first_mask' in g:
        mask = torch.from_numpy(g['first_mask'].astype(np.float32))[None, None].to(device)
    return frames, mask, objs, out_size


def integration_snippet():
    """The python code block of INTEGRATION.md section 2 (binding the C ABI directly), verbatim."""
    import re
    doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    sec = doc.split('## 2. Binding the C ABI directly')[1].split('\n## ')[0]
    return re.search(r'```python\n(.*?)```', sec, re.S).group(1)


def lstt_last_of(engine):
    """Last LSTT/GPM layer output after its decoder norm, [N, C] (AOT) / [N, 2C] (DeAOT) -- what the reference keeps
    in curr_lstt_output[0][-1] (aot_engine.py:340-354); first object group."""
    e0 = engine.aot_engines[0] if hasattr(engine, 'aot_engines') else engine
    if hasattr(e0, 'curr_lstt_output'):                       # oracle / reference layout [N, 1, C]
        return e0.curr_lstt_output[0][-1][:, 0]
    return e0.lstt_last()


def unpack_gapmask(g, t, shape):
    bits = np.unpackbits(g['gapmask_%d' % t])[:shape[0] * shape[1]]
    return bits.reshape(shape).astype(bool)


from fp64_ties import FP64_GAP_BINS, add_counts, classify_flips, load_fp64_ties  # noqa: E402,F401  (pure numpy; bench.py's J&F leg uses it too)


def check_masks(pred, g, t, what):
    """argmax mask ids must equal the reference's except on the reference's own near-tie pixels."""
    ref = g['masks'][t - 1]
    pred = np.asarray(pred)
    bad = pred != ref
    tie = unpack_gapmask(g, t, ref.shape)
    hard = int((bad & ~tie).sum())
    assert hard == 0, '%s frame %d: %d mask pixels differ outside near-ties (total diff %d)' % (what, t, hard, int(bad.sum()))
    assert int(bad.sum()) <= max(8, int(tie.sum())), '%s frame %d: too many tie flips %d' % (what, t, int(bad.sum()))
    return int(bad.sum())


def run_teacher_forced(engine, frames, mask, objs, out_size, g, keep, to_dev=lambda x: x, extra=None, sub=None,
                       label_fn=None):
    """demo loop (tools/demo.py:187-235) with the GOLDEN mask fed back into memory at every frame, so frame t is
    compared on identical history.  Returns {t: (logits4 [no,h,w], mask uint8 [H,W])}.  `extra` (dict) receives, for the
    kept frames, 'lstt_last_<t>' and -- with `sub` -- the merged output-size logits subsampled by `sub`.  label_fn(logit)
    -> [1,1,H,W] label map replaces torch's softmax -> argmax (the GPU tests pass aot_hip.fuse_probs, what bench.py runs)."""
    out = {}
    engine.restart_engine()
    with torch.no_grad():
        engine.add_reference_frame(to_dev(frames[0]), to_dev(mask), objs, frame_step=0)
        for t in range(1, len(frames)):
            engine.match_propogate_one_frame(to_dev(frames[t]))
            logit = engine.decode_current_logits(out_size)
            lab = torch.argmax(torch.softmax(logit, dim=1), dim=1, keepdim=True) if label_fn is None else label_fn(logit)
            l4 = engine.pred_id_logits if hasattr(engine, 'pred_id_logits') else engine.aot_engines[0].pred_id_logits
            out[t] = (l4[0].detach().float().cpu().numpy() if t in keep else None, lab[0, 0].to(torch.uint8).cpu().numpy())
            if extra is not None and t in keep:
                extra['lstt_last_%d' % t] = lstt_last_of(engine).detach().float().cpu().numpy()
                if sub:
                    extra['merged_%d' % t] = logit[0, :, ::sub, ::sub].detach().float().cpu().numpy()
            fb = torch.from_numpy(g['masks'][t - 1].astype(np.float32)).view(1, 1, *out_size)
            fb = F.interpolate(to_dev(fb), size=engine.input_size_2d, mode='nearest')
            engine.update_memory(fb)
    return out


def mha_knob_inputs():
    """Inputs of tests/golden/mha_knobs.npz (make_golden.make_mha_knobs): same seeded CPU generator."""
    g = torch.Generator().manual_seed(4242)
    Tq, Tk, C, H = 96, 1000, 256, 8
    Q = torch.randn(Tq, 1, C, generator=g) * 2.0
    K = torch.randn(Tk, 1, C, generator=g)
    V = torch.randn(Tk, 1, C, generator=g)
    return Q, K, V, H


MHA_KNOB_CASES = {'topk50': dict(top_k=50), 'topk1': dict(top_k=1), 'ratio4': dict(max_mem_len_ratio=4.),
                  'ratio4_topk200': dict(max_mem_len_ratio=4., top_k=200), 'dense': dict()}


def gp_knob_inputs():
    """Inputs of tests/golden/gp_knobs.npz (make_golden.make_gp_knobs): one 128-wide head, value / gate width 1024, a
    13x11 query map against 900 memory tokens."""
    g = torch.Generator().manual_seed(777)
    h, w, Tk = 13, 11, 900
    Q = torch.randn(h * w, 1, 128, generator=g) * 3.0
    K = torch.randn(Tk, 1, 128, generator=g)
    V = torch.randn(Tk, 1, 1024, generator=g)
    U = torch.randn(h * w, 1, 1024, generator=g)
    return Q, K, V, U, (h, w)


GP_KNOB_CASES = {'topk40': dict(top_k=40), 'topk1': dict(top_k=1), 'ratio3': dict(max_mem_len_ratio=3.),
                 'ratio3_topk300': dict(max_mem_len_ratio=3., top_k=300), 'dense': dict()}


def gp_knob_state(module_state_dict):
    """The keyed synthetic weights make_gp_knobs loaded into the reference module (key names prefixed 'gp_knobs.')."""
    from utils.synth import synth_state_dict
    keyed = synth_state_dict({'gp_knobs.' + k: v for k, v in module_state_dict.items()})
    return {k[len('gp_knobs.'):]: v for k, v in keyed.items()}


# training-side slice: loss cases of tests/golden/training_losses.npz (make_golden.make_training)
LOSS_CASES = {
    'ce_topk_start': dict(kind='ce', top_k=0.15, mining_steps=1000, step=0, seed=1),
    'ce_topk_mid': dict(kind='ce', top_k=0.15, mining_steps=1000, step=400, seed=2),
    'ce_topk_end': dict(kind='ce', top_k=0.15, mining_steps=1000, step=5000, seed=3),
    'ce_mean': dict(kind='ce', top_k=None, mining_steps=1000, step=10, seed=4),
    'jaccard': dict(kind='jac', step=0, seed=5),
    'jaccard_missing_classes': dict(kind='jac', step=0, seed=6),
}


def loss_case_inputs(name):
    """Three samples with 4 / 11 / 2 classes (objects + background, as aot_engine.py:406-412 slices them) of 37x53 pixels;
    labels carry an ignore band (255); 'missing' leaves some classes without pixels."""
    c = LOSS_CASES[name]
    g = torch.Generator().manual_seed(9000 + c['seed'])
    logits, labels = [], []
    for C in (4, 11, 2):
        logits.append(torch.randn(1, C, 37, 53, generator=g) * 2.0)
        hi = C if 'missing' not in name else max(2, C // 2)
        lab = torch.randint(0, hi, (1, 37, 53), generator=g).float()
        lab[:, :3, :] = 255.
        lab[:, 20:22, 10:30] = 255.
        labels.append(lab)
    return logits, labels


# ---- training-step forward (aot_engine.py:33-108): reference goldens in tests/golden/train_forward.npz ----------------
TRAIN_CFG = dict(TRAIN_TOTAL_STEPS=100000, TRAIN_TOP_K_PERCENT_PIXELS=0.15, TRAIN_HARD_MINING_RATIO=0.5,
                 TRAIN_AUX_LOSS_WEIGHT=1.0, TRAIN_AUX_LOSS_RATIO=1.0)      # the reference's defaults (configs/default.py:37-68)
TRAIN_FWD_CASES = {
    # ground-truth feedback (the default of the first training half), mid-way through the hard-example annealing
    'tf_aott': dict(model='aott', size=(129, 161), frames=4, objs=(3, 1), step=30000),
    # sequential-training half: the prediction is fed back; second frame memorises its own mask (TRAIN_ENABLE_PREV_FRAME)
    'tf_aott_prev': dict(model='aott', size=(129, 161), frames=5, objs=(2, 4), step=70000, use_prev_pred=True,
                         enable_prev_frame=True),
    # identities shuffled per sample (trainer.py:457), auxiliary loss faded out, top-k at its final share
    'tf_aott_shuffle': dict(model='aott', size=(113, 145), frames=4, objs=(3, 2), step=120000, shuffle=True),
    # DeAOT; probabilities fed back (MODEL_USE_PREV_PROB) through the dense identity convolution, shuffled as well
    'tf_deaott_prob': dict(model='deaott', size=(129, 161), frames=4, objs=(2, 3), step=0, use_prev_pred=True,
                           use_prev_prob=True, shuffle=True),
    # the model BASELINE config 5 trains (R50-DeAOTL: ResNet-50 trunk, three GPM layers), second self-memorising frame, shuffled
    'tf_r50_deaotl': dict(model='r50_deaotl', size=(129, 161), frames=5, objs=(3, 2), step=50000, enable_prev_frame=True,
                          shuffle=True),
    # BASELINE config 3's model (Swin-B trunk: windows padded 32 x 40 -> 35 x 42, shifted-window masks; MODEL_ALIGN_CORNERS = False: sizes in multiples of 16), prediction feedback
    'tf_swinb_deaotl': dict(model='swinb_deaotl', size=(128, 160), frames=4, objs=(2, 3), step=20000, use_prev_pred=True,
                            shuffle=True),
}


# gradient goldens (tests/golden/train_grads.npz, make_golden.make_train_grads): the cases, the parameters stored in full, and
# the fixed subsample of every other parameter's gradient
TRAIN_GRAD_CASES = ('tf_aott', 'tf_deaott_prob', 'tf_r50_deaotl', 'tf_swinb_deaotl')
TRAIN_GRAD_FULL = ('patch_wise_id_bank.bias', 'decoder.conv_out.weight', 'decoder.conv_out.bias', 'LSTT.layers.0.norm1.weight',
                   'encoder_projector.bias')


def grad_sample_index(n):
    """64 fixed positions of a flattened gradient of n entries (all of it when n <= 64)."""
    if n <= 64:
        return torch.arange(n)
    return (torch.arange(64, dtype=torch.float64) * (n - 1) / 63.0).round().long()


def check_grads_against_golden(case, grads, g, rel=5e-3):
    """Holds {parameter name: gradient} to the reference's gradient fixture `g` (train_grads.npz) of `case`: every parameter
    the reference gives a gradient has one, L2 norm within 0.2 %, the 64 sampled entries within rel x the rms entry (+ 0.2 % of
    the largest), the few small tensors stored in full elementwise.  Returns the worst sampled error in units of the rms."""
    names = [str(n) for n in g[case + '.names']]
    worst = 0.0
    for i, k in enumerate(names):
        gr = grads.get(k)
        assert gr is not None, 'no gradient reached %s' % k
        flat = gr.detach().double().flatten().cpu()
        ref_norm = float(g[case + '.norm'][i])
        tol = 2e-3 * ref_norm + 1e-6
        assert abs(float(flat.norm()) - ref_norm) <= tol, '%s: |grad| %g vs %g' % (k, float(flat.norm()), ref_norm)
        idx = grad_sample_index(flat.numel())
        got = flat[idx].numpy()
        ref = g[case + '.sample'][i][:got.size]
        scale = ref_norm / max(1.0, flat.numel()) ** 0.5            # rms entry of the reference gradient
        err = float(np.abs(got - ref).max())
        assert err <= rel * scale + 2e-3 * float(np.abs(ref).max()) + 1e-7, '%s: sampled entries differ by %g (rms %g)' % (k, err, scale)
        if ref_norm > 1e-6:                                         # (a key bias has gradient 0 up to rounding: softmax ignores it)
            worst = max(worst, err / scale)
        if k in TRAIN_GRAD_FULL:
            full = g['%s.full.%s' % (case, k)]
            np.testing.assert_allclose(gr.detach().cpu().numpy(), full, rtol=rel, atol=rel * scale + 1e-7)
    extra = [k for k, v in grads.items() if v is not None and k not in names]
    assert not extra, 'gradients on parameters the reference leaves without one: %s' % extra[:5]
    return worst


def train_batch(name):
    """The batch of a TRAIN_FWD_CASES entry, rebuilt from seeds: sample b is synthetic clip 30+b with objs[b] objects; the
    label of frame t is the first-frame label moved with the clip's motion (synth_clip rolls by (2t, 3t)); one sample
    carries an ignore band.  Returns (all_frames [T*bs,3,H,W], all_masks [T*bs,1,H,W], obj_nums, id permutations | None)
    in the trainer's time-major order (trainer.py:452-455)."""
    from utils.synth import synth_clip
    c = TRAIN_FWD_CASES[name]
    T, bs = c['frames'], len(c['objs'])
    frames, masks = [], []
    for b, n in enumerate(c['objs']):
        f, m, _, _ = synth_clip(30 + b, T, c['size'], c['size'], n)
        frames.append(f)
        ms = [torch.roll(m, shifts=(2 * t, 3 * t), dims=(2, 3)) for t in range(T)]
        if b == 1:
            ms[2] = ms[2].clone()
            ms[2][:, :, 5:9, :] = 255.
        masks.append(ms)
    all_frames = torch.cat([frames[b][t] for t in range(T) for b in range(bs)], 0)
    all_masks = torch.cat([masks[b][t] for t in range(T) for b in range(bs)], 0)
    perms = None
    if c.get('shuffle'):
        g = torch.Generator().manual_seed(77)
        perms = [torch.cat([torch.zeros(1, dtype=torch.long), 1 + torch.randperm(10, generator=g)]) for _ in range(bs)]
    return all_frames, all_masks, list(c['objs']), perms


# ---- evaluator loop (evaluator.py:209-505): reference goldens in tests/golden/evaluator_loop.npz ----------------------
EVAL_LOOP_CASES = {'single': dict(flip=False, ms=(1,)), 'tta': dict(flip=True, ms=(1.3, 1.0))}
EVAL_LOOP_OBJ_IDX = [0, 5, 9, 12]          # dataset object ids of the dense ids 0..3


def evaluator_scenario():
    """Four 96x150 frames of smoothed noise sliding by (1, 2) px per frame, two objects labelled in frame 0 and a third
    injected at frame 2 (dense ids 1..3; the dataset's own ids are EVAL_LOOP_OBJ_IDX).  Returns (frames: list of float32
    [H,W,3] in 0..255, labels {frame: [H,W] dense ids}, obj_nums {frame: objects known BEFORE that frame's label})."""
    rs = np.random.RandomState(3)
    H, W = 96, 150
    base = rs.rand(H + 8, W + 8, 3).astype(np.float32)
    k = np.ones(5, np.float32) / 5
    for ax in (0, 1):       # smooth the noise so the cubic resize is well conditioned
        base = np.apply_along_axis(lambda v: np.convolve(v, k, mode='same'), ax, base)
    base = (base - base.min()) / (base.max() - base.min()) * 255
    frames = [np.ascontiguousarray(base[t:t + H, 2 * t:2 * t + W]).astype(np.float32) for t in range(4)]
    lab0 = np.zeros((H, W), np.float32)
    lab0[20:60, 30:80] = 1
    lab0[50:90, 90:140] = 2
    lab2 = np.zeros((H, W), np.float32)
    lab2[5:25, 100:140] = 3
    return frames, {0: lab0, 2: lab2}, {0: 2, 2: 3}


# ---- objects that appear mid-clip and open a second object group (aot_engine.py:584-609; evaluator.py:362-393) -----------
NEWGROUP_CASE = dict(model='aott', gap=2, frames=6, in_size=(129, 161), out_size=(128, 160), clip=13, first=9, total=13, inject=2)


def newgroup_clip(device='cpu'):
    """Synthetic clip 13 with 13 rectangles: objects 1..9 are labelled in frame 0, objects 10..13 are injected at frame 2
    (their rectangles moved with the clip's motion).  Returns (frames, first_mask [1,1,H,W], new_label [1,1,oh,ow] at output
    size -- zero where nothing is injected, as the evaluator's `new_obj_label`)."""
    from utils.synth import synth_clip
    c = NEWGROUP_CASE
    frames, mask, _, _ = synth_clip(c['clip'], c['frames'], c['in_size'], c['out_size'], c['total'], device=device)
    first = torch.where(mask <= c['first'], mask, torch.zeros_like(mask))
    late = torch.where(mask > c['first'], mask, torch.zeros_like(mask))
    late = torch.roll(late, shifts=(2 * c['inject'], 3 * c['inject']), dims=(2, 3))
    new_label = F.interpolate(late, size=c['out_size'], mode='nearest')
    return frames, first, new_label


def run_newgroup(engine, frames, first, new_label, feedback, to_dev=lambda x: x):
    """The evaluator's per-frame calls (evaluator.py:318-408, one augmentation) with new objects at frame `inject`;
    feedback(t, logits) -> the label map [1,1,oh,ow] to memorise (the test feeds the golden masks, the golden run its own
    argmax).  Returns the per-frame output-size logits."""
    c = NEWGROUP_CASE
    out = []
    engine.restart_engine()
    resize = lambda lab: F.interpolate(lab, size=engine.input_size_2d, mode='nearest')
    with torch.no_grad():
        engine.add_reference_frame(to_dev(frames[0]), to_dev(first), [c['first']], frame_step=0)
        for t in range(1, len(frames)):
            engine.match_propogate_one_frame(to_dev(frames[t]))
            logit = engine.decode_current_logits(c['out_size'])
            out.append(logit)
            label = to_dev(feedback(t, logit))
            if t == c['inject']:
                nl = to_dev(new_label)
                keep = (nl == 0).float()
                label = label * keep + nl * (1 - keep)
                engine.add_reference_frame(to_dev(frames[t]), resize(label), [int(label.max().item())], frame_step=t)
                engine.decode_current_logits(c['out_size'])
            engine.update_memory(resize(label))
    return out


# ---- memory schedule options of the engine (aot_engine.py:307-338: short_term_mem_skip, skip_long_term_update) -----------
MEMSCHED_CASES = {
    'aott': dict(model='aott', frames=8, in_size=(129, 161), out_size=(128, 160), num_obj=3, clip=14, gap=2, skip=2),
    'deaott': dict(model='deaott', frames=8, in_size=(129, 161), out_size=(128, 160), num_obj=2, clip=15, gap=2, skip=3),
}


def memsched_skip_long(t):
    """Frames whose long-term update is suppressed by the caller (skip_long_term_update=True)."""
    return t % 3 == 0


def run_memsched(engine, frames, mask, objs, out_size, feedback, to_dev=lambda x: x):
    """Demo loop with skip_long_term_update on every third frame; feedback(t, logits) -> label map [1,1,oh,ow]."""
    out = []
    engine.restart_engine()
    with torch.no_grad():
        engine.add_reference_frame(to_dev(frames[0]), to_dev(mask), objs, frame_step=0)
        for t in range(1, len(frames)):
            engine.match_propogate_one_frame(to_dev(frames[t]))
            logit = engine.decode_current_logits(out_size)
            out.append(logit)
            fb = F.interpolate(to_dev(feedback(t, logit)), size=engine.input_size_2d, mode='nearest')
            engine.update_memory(fb, skip_long_term_update=memsched_skip_long(t))
    return out
