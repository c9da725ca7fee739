"""Label helpers on the hot path (reference utils/image.py:69-74)."""
import torch


def one_hot_mask(mask, cls_num):
    """[B,1,H,W] (or [B,H,W]) label ids -> [B,cls_num+1,H,W] float one-hot; ids above
    ``cls_num`` (or non-integer values) give an all-zero column, as in the reference."""
    if mask.dim() == 3:
        mask = mask.unsqueeze(1)
    ids = torch.arange(0, cls_num + 1, device=mask.device).view(1, -1, 1, 1)
    return (mask == ids).float()
