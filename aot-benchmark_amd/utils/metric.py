"""Region similarity J and boundary accuracy F between two label maps (the DAVIS measures the reference's README
defers to external toolkits for; `pytorch_iou`, reference utils/metric.py:4-36, is the only in-repo J-like
metric).  Used to report "J&F vs reference": masks of this engine scored against the reference's own masks."""
import torch
import torch.nn.functional as F


def _boundary(mask):
    """1-pixel inner boundary of a boolean [H,W] mask (pixels whose right/down/diag neighbour differs)."""
    m = mask.float()
    e = torch.zeros_like(m)
    e[:, :-1] += (m[:, :-1] != m[:, 1:]).float()
    e[:-1, :] += (m[:-1, :] != m[1:, :]).float()
    e[:-1, :-1] += (m[:-1, :-1] != m[1:, 1:]).float()
    return (e > 0) & mask


def _dilate(b, r):
    if r <= 0:
        return b
    return F.max_pool2d(b.float()[None, None], 2 * r + 1, 1, r)[0, 0] > 0


def jf_per_object(pred, ref, num_obj, bound_th=0.008):
    """pred, ref: integer label maps [H,W]; returns (J, F) averaged over objects 1..num_obj present in either map.
    F follows the DAVIS definition: boundary precision/recall with a tolerance of bound_th * image diagonal."""
    H, W = ref.shape[-2:]
    r = max(1, int(round(bound_th * (H * H + W * W) ** 0.5)))
    js, fs = [], []
    for o in range(1, num_obj + 1):
        p, g = pred == o, ref == o
        if not (p.any() or g.any()):
            continue
        inter, union = (p & g).sum().item(), (p | g).sum().item()
        js.append(inter / union if union else 1.0)
        bp, bg = _boundary(p), _boundary(g)
        if not bp.any() and not bg.any():
            fs.append(1.0)
            continue
        prec = (bp & _dilate(bg, r)).sum().item() / max(1, bp.sum().item())
        rec = (bg & _dilate(bp, r)).sum().item() / max(1, bg.sum().item())
        fs.append(0.0 if prec + rec == 0 else 2 * prec * rec / (prec + rec))
    if not js:
        return 1.0, 1.0
    return sum(js) / len(js), sum(fs) / len(fs)
