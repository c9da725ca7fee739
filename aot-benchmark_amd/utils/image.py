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


def restrict_size(h, w, max_short_edge=None, max_long_edge=800, scale=1.0, align_corners=True, max_stride=16):
    """Network input size of an h x w frame: MultiRestrictSize's arithmetic, dataloaders/video_transforms.py:612-653
    (short/long edge caps, the multi-scale factor, then rounding to k*stride(+1 with align_corners))."""
    sc = 1.
    if max_short_edge is not None:
        short_edge = w if h > w else h
        if short_edge > max_short_edge:
            sc *= float(max_short_edge) / short_edge
    new_h, new_w = sc * h, sc * w
    sc = 1.
    if max_long_edge is not None:
        long_edge = new_h if new_h > new_w else new_w
        if long_edge > max_long_edge:
            sc *= float(max_long_edge) / long_edge
    new_h, new_w = sc * new_h, sc * new_w
    new_h = int(new_h * scale)
    new_w = int(new_w * scale)
    if align_corners:
        if (new_h - 1) % max_stride != 0:
            new_h = int(np.around((new_h - 1) / max_stride) * max_stride + 1)
        if (new_w - 1) % max_stride != 0:
            new_w = int(np.around((new_w - 1) / max_stride) * max_stride + 1)
    else:
        if new_h % max_stride != 0:
            new_h = int(np.around(new_h / max_stride) * max_stride)
        if new_w % max_stride != 0:
            new_w = int(np.around(new_w / max_stride) * max_stride)
    return new_h, new_w
