"""Drives the *real* reference (/root/reference, build container only) on in-memory
tensors.  Used by make_golden.py; never imported on the GPU box.

Follows SURVEY.md Appendix C: model configs are instantiated without the engine
config (its init_dir() creates directories), and -- for AOT models only -- the shipped
``MultiheadLocalAttentionV3`` (broken: attention.py:527-532 adds [n,H,hw,d] to
[hw,n,C]) is replaced by ``MultiheadLocalAttentionV2(enable_corr=False)``, which has
the same parameters and is the semantics the CUDA correlation extension implements.
"""
import importlib
import sys

import torch
import torch.nn.functional as F

REF = '/root/reference'


def _enter():
    sys.dont_write_bytecode = True
    # the reference's top-level packages are called networks/configs/utils, like ours: evict ours
    for name in list(sys.modules):
        if name.split('.')[0] in ('networks', 'configs', 'utils', 'dataloaders'):
            del sys.modules[name]
    # our package dir holds REGULAR packages with the same names; the reference's are namespace packages (no
    # __init__.py), which lose against a regular package anywhere on sys.path -- so take ours off the path
    global _saved
    _saved = [p for p in sys.path if p.rstrip('/').endswith('aot-benchmark_amd')]
    sys.path[:] = [p for p in sys.path if p not in _saved]
    sys.path.insert(0, REF)


_saved = []


def _leave():
    sys.path.remove(REF)
    sys.path[:0] = _saved
    for name in list(sys.modules):
        if name.split('.')[0] in ('networks', 'configs', 'utils', 'dataloaders'):
            del sys.modules[name]


def build_reference(model_name, gap=None):
    """Returns (net, engine_factory, cfg) built from the reference's own code."""
    _enter()
    try:
        cfg = importlib.import_module('configs.models.' + model_name).ModelConfig()
        for k, v in dict(TRAIN_ENCODER_FREEZE_AT=2, TRAIN_LSTT_EMB_DROPOUT=0., TRAIN_LSTT_ID_DROPOUT=0.,
                         TRAIN_LSTT_DROPPATH=0.1, TRAIN_LSTT_DROPPATH_SCALING=False,
                         TRAIN_LSTT_DROPPATH_LST=False, TRAIN_LSTT_LT_DROPOUT=0.,
                         TRAIN_LSTT_ST_DROPOUT=0.).items():
            setattr(cfg, k, v)
        import networks.layers.transformer as T
        from networks.layers.attention import MultiheadLocalAttentionV2

        def v2(d_model, nhead, dilation=1, use_linear=False, dropout=0.):
            return MultiheadLocalAttentionV2(d_model, nhead, dilation=dilation, use_linear=use_linear,
                                             dropout=dropout, enable_corr=False)
        T.MultiheadLocalAttentionV3 = v2
        from networks.models import build_vos_model
        from networks.engines import build_engine
        net = build_vos_model(cfg.MODEL_VOS, cfg).eval()

        def make_engine():
            return build_engine(cfg.MODEL_ENGINE, phase='eval', aot_model=net, gpu_id=-1,
                                long_term_mem_gap=cfg.TEST_LONG_TERM_MEM_GAP if gap is None else gap)
        return net, make_engine, cfg
    finally:
        _leave()


def run_reference_clip(engine, frames, first_mask, obj_nums, output_size, teacher_masks=None, keep=None):
    """tools/demo.py:187-235 on tensors; returns per-frame dicts (mask, logits4, and
    per-layer LSTT outputs of the first engine)."""
    out = []
    engine.restart_engine()
    with torch.no_grad():
        engine.add_reference_frame(frames[0], first_mask, frame_step=0, obj_nums=obj_nums)
        for t in range(1, len(frames)):
            engine.match_propogate_one_frame(frames[t])
            logit = engine.decode_current_logits(output_size)
            prob = torch.softmax(logit, dim=1)
            label = torch.argmax(prob, dim=1, keepdim=True).float()
            e0 = engine.aot_engines[0]
            rec = {'mask': label[0, 0].to(torch.uint8)}
            top2 = torch.topk(logit[0], 2, dim=0)[0]
            rec['gap'] = (top2[0] - top2[1]).clone()
            if keep is None or t in keep:          # heavy tensors only for the frames the caller stores
                rec.update({'logits4': e0.pred_id_logits.clone(), 'logits': logit.clone(),
                            'lstt': [x.clone() for x in e0.curr_lstt_output[0]]})
            fb = label if teacher_masks is None else teacher_masks[t - 1].view(1, 1, *output_size).float()
            fb = F.interpolate(fb, size=engine.input_size_2d, mode='nearest')
            engine.update_memory(fb)
            out.append(rec)
    return out
