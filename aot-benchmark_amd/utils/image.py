"""Label helpers on the hot path (reference utils/image.py:69-74) and the evaluator's size rule."""
import numpy as np
import torch


def one_hot_mask(mask, cls_num):
    """[B,1,H,W] (or [B,H,W]) label ids -> [B,cls_num+1,H,W] float one-hot; ids above
    ``cls_num`` (or non-integer values) give an all-zero column, as in the reference."""
    if mask.dim() == 3:
        mask = mask.unsqueeze(1)
    ids = torch.arange(0, cls_num + 1, device=mask.device).view(1, -1, 1, 1)
    return (mask == ids).float()


def _snap(n, stride, plus_one):
    """n -> nearest k*stride (+1): numpy's round-half-to-even on the quotient, as the reference uses np.around."""
    off = 1 if plus_one else 0
    if (n - off) % stride == 0:
        return n
    return int(np.around((n - off) / stride) * stride + off)


def restrict_size(h, w, max_short_edge=None, max_long_edge=800, scale=1.0, align_corners=True, max_stride=16):
    """Network input size of an h x w frame (the rule of MultiRestrictSize, dataloaders/video_transforms.py:612-653):
    cap the short edge, then the long edge (both by uniform float scaling), apply the multi-scale factor with truncation,
    then snap each side to k*stride (+1 when the model aligns corners).  Pinned on the reference class by
    tests/golden/transforms.json."""
    fh, fw = float(h), float(w)
    if max_short_edge is not None and min(h, w) > max_short_edge:
        r = float(max_short_edge) / min(h, w)
        fh, fw = r * fh, r * fw
    if max_long_edge is not None and max(fh, fw) > max_long_edge:
        r = float(max_long_edge) / max(fh, fw)
        fh, fw = r * fh, r * fw
    return (_snap(int(fh * scale), max_stride, align_corners), _snap(int(fw * scale), max_stride, align_corners))
