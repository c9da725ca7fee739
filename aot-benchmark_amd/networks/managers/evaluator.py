"""Per-sequence evaluation loop (reference networks/managers/evaluator.py:236-446) on in-memory frames.

The reference's ``Evaluator`` couples this loop to its dataset classes, PNG writer and result zipping; what sits on
either side of the engine per frame -- the multi-scale / flip test-time augmentation, the probability fusion, the
new-object merge and the label feedback to every augmentation's engine -- is restated here around device kernels
(``aot_hip.preprocess`` / ``fuse_probs`` / ``label_resize``), with frames and labels already resident on the GPU.

Config keys read (same names as configs/default.py of the reference): TEST_FLIP, TEST_MULTISCALE, TEST_MAX_SHORT_EDGE,
TEST_MAX_LONG_EDGE, TEST_LONG_TERM_MEM_GAP, TEST_SHORT_TERM_MEM_SKIP, MODEL_ALIGN_CORNERS, MODEL_ENGINE, MODEL_USE_PREV_PROB.
"""
import os

import torch

import aot_hip
from networks.engines import build_engine
from utils.image import restrict_size, save_mask


class SequenceEvaluator:
    def __init__(self, cfg, model, gpu_id=0):
        # MODEL_USE_PREV_PROB: every augmentation's engine gets its own class probabilities back instead of its label map
        # (evaluator.py:409-425 -- as shipped that branch stops on an undefined name, `current_prob`; what it sets out to do
        # is done here).  One object group only (see AOT.id_emb_from_mask).
        self.use_prev_prob = bool(getattr(cfg, 'MODEL_USE_PREV_PROB', False))
        self.cfg = cfg
        self.model = model
        self.gpu = gpu_id
        self.engines = []

    def augmentations(self, h, w):
        """[(in_h, in_w, flip)] in the order MultiRestrictSize emits its samples (video_transforms.py:609-682)."""
        cfg = self.cfg
        out = []
        for scale in cfg.TEST_MULTISCALE:
            nh, nw = restrict_size(h, w, cfg.TEST_MAX_SHORT_EDGE, cfg.TEST_MAX_LONG_EDGE, scale, cfg.MODEL_ALIGN_CORNERS)
            out.append((nh, nw, False))
            if cfg.TEST_FLIP:
                out.append((nh, nw, True))
        return out

    def _engine(self, i):
        while len(self.engines) <= i:                                    # evaluator.py:272-283
            e = build_engine(self.cfg.MODEL_ENGINE, phase='eval', aot_model=self.model, gpu_id=self.gpu,
                             long_term_mem_gap=self.cfg.TEST_LONG_TERM_MEM_GAP,
                             short_term_mem_skip=self.cfg.TEST_SHORT_TERM_MEM_SKIP)
            e.eval()
            self.engines.append(e)
        return self.engines[i]

    @torch.no_grad()
    def run(self, frames, labels, obj_nums, save_dir=None, names=None, obj_idx=None):
        """frames: list of [H, W, 3] uint8/float32 device images (values 0..255, the dataset's channel order);
        labels: {frame_idx: [H, W] label map} -- frame 0 is required, later entries inject new objects
        (evaluator.py:336-338,362-392); obj_nums: {frame_idx: int} objects annotated so far at that frame.
        Returns the list of predicted label maps [H, W] (float) for frames 1.. .
        save_dir: the predictions are also written as palette PNGs `<save_dir>/<names[t]>.png` once the sequence is
        done (outside the per-frame loop, as the reference does: evaluator.py:448-466,500-505); obj_idx = the dataset's
        object ids of the dense ids 0..n (the reference's squeeze index), or None."""
        H, W = frames[0].shape[:2]
        augs = self.augmentations(H, W)
        for e in self.engines:
            e.restart_engine()
        preds = []
        for t, img in enumerate(frames):
            inputs = [aot_hip.preprocess(img, nh, nw, flip) for (nh, nw, flip) in augs]
            label = labels.get(t)
            if label is not None:
                label = label.to(torch.float32).reshape(1, 1, H, W)
            if t == 0:
                for i, (nh, nw, flip) in enumerate(augs):
                    # the flipped sample carries the flipped label (video_transforms.py:669-680), resized by nearest (:309-312)
                    lab = aot_hip.label_resize(label, nh, nw, flip)
                    self._engine(i).add_reference_frame(inputs[i], lab, frame_step=0, obj_nums=[int(obj_nums[0])])
                continue
            logits = []
            for i in range(len(augs)):
                e = self._engine(i)
                e.match_propogate_one_frame(inputs[i])
                logits.append(e.decode_current_logits((H, W)))
            nc = max(l.shape[1] for l in logits)
            if any(l.shape[1] != nc for l in logits):
                raise RuntimeError('augmentation engines disagree on the number of logit channels')
            fused, augl, _ = aot_hip.fuse_probs(torch.cat(logits, 0), [f for (_, _, f) in augs], new_label=label)
            preds.append(fused[0, 0])
            if label is not None:                                           # new objects appear in this frame (:362-392)
                new_obj_nums = [int(fused.max().item())]
                for i, (nh, nw, flip) in enumerate(augs):
                    lab = aot_hip.label_resize(augl[i], nh, nw, flip)
                    e = self._engine(i)
                    e.add_reference_frame(inputs[i], lab, obj_nums=new_obj_nums, frame_step=t)
                    e.decode_current_logits((H, W))
                    e.update_memory(lab)
            elif self.use_prev_prob:
                for i, (nh, nw, flip) in enumerate(augs):                   # (:409-425)
                    # the engine's own softmax, in the orientation it decoded in, nearest-resized plane by plane
                    _, _, prob = aot_hip.fuse_probs(logits[i], [False], want_aug_labels=False, want_prob=True)
                    self._engine(i).update_memory(torch.cat([aot_hip.label_resize(prob[0, c], nh, nw)
                                                             for c in range(prob.shape[1])], 1))
            else:
                for i, (nh, nw, flip) in enumerate(augs):                   # (:394-408)
                    self._engine(i).update_memory(aot_hip.label_resize(augl[i], nh, nw, flip))
        if save_dir is not None:
            os.makedirs(save_dir, exist_ok=True)
            writers = [save_mask(p, os.path.join(save_dir, '%s.png' % (names[t] if names is not None else '%05d' % t)),
                                 obj_idx) for t, p in enumerate(preds, start=1)]
            for w in writers:
                w.join()
        return preds
