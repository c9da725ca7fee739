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


def pytorch_iou(pred, target, obj_num, epsilon=1e-6):
    """The trainer's running IoU (reference utils/metric.py:4-36), on the tensors' device.  pred / target [bs, H, W] label
    maps, obj_num [bs]: per sample the mean over objects 1..n of (|p & t| + eps) / (|p | t| + eps); samples without objects
    are skipped, a batch without any object scores 1.
    The trainer itself passes [bs, 1, H, W] maps (trainer.py:507-509); the reference's reduction over dims (1, 2) then runs
    over (objects, rows) and the ratio is averaged over image COLUMNS.  That is what the logged number is, so it is kept:
    the reduction below is over dims (1, 2) of the same broadcast shape for either rank."""
    per_sample = []
    for b in range(pred.shape[0]):
        n = int(obj_num[b])
        if n == 0:
            continue
        ids = torch.arange(1, n + 1, device=pred.device).view(-1, 1, 1)
        p = pred[b].unsqueeze(0) == ids                   # [n, H, W], or [1, n, H, W] for 4-D input
        t = target[b].unsqueeze(0) == ids
        inter = (p & t).sum((1, 2)).float()
        union = (p | t).sum((1, 2)).float()
        per_sample.append(((inter + epsilon) / (union + epsilon)).mean())
    if per_sample:
        return torch.stack(per_sample).mean()
    return torch.ones(1, device=pred.device)
