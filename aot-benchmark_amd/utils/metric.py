"""Region similarity J and boundary accuracy F between two label maps (the DAVIS measures the reference's README
defers to external toolkits for; `pytorch_iou`, reference utils/metric.py:4-36, is the only in-repo J-like
metric).  Used to report "J&F vs reference": masks of this engine scored against the reference's own masks."""
import torch
import torch.nn.functional as F


def _boundary(mask):
    """1-pixel inner boundary of boolean [..., H, W] masks (pixels whose right / down / diagonal neighbour differs)."""
    e = torch.zeros_like(mask)
    e[..., :, :-1] |= mask[..., :, :-1] != mask[..., :, 1:]
    e[..., :-1, :] |= mask[..., :-1, :] != mask[..., 1:, :]
    e[..., :-1, :-1] |= mask[..., :-1, :-1] != mask[..., 1:, 1:]
    return e & mask


def _dilate(b, r):
    """Dilation by a (2r+1) x (2r+1) square as two 1-D max filters; b: boolean [..., H, W]."""
    if r <= 0:
        return b
    x = b.to(torch.float32).reshape(-1, 1, *b.shape[-2:])
    x = F.max_pool2d(x, (2 * r + 1, 1), 1, (r, 0))
    x = F.max_pool2d(x, (1, 2 * r + 1), 1, (0, r))
    return x.reshape(b.shape) > 0


def jf_per_object(pred, ref, num_obj, bound_th=0.008):
    """pred, ref: integer label maps [H,W]; returns (J, F) averaged over objects 1..num_obj present in either map.
    F follows the DAVIS definition: boundary precision/recall with a tolerance of bound_th * image diagonal.
    All objects are scored in one pass over [num_obj, H, W] stacks."""
    H, W = ref.shape[-2:]
    r = max(1, int(round(bound_th * (H * H + W * W) ** 0.5)))
    if num_obj < 1:
        return 1.0, 1.0
    ids = torch.arange(1, num_obj + 1, device=ref.device).view(-1, 1, 1)
    p, g = pred.reshape(1, H, W) == ids, ref.reshape(1, H, W) == ids
    present = (p | g).flatten(1).any(1)
    inter, union = (p & g).flatten(1).sum(1), (p | g).flatten(1).sum(1)
    bp, bg = _boundary(p), _boundary(g)
    wide = _dilate(torch.cat([bg, bp], 0), r)
    hit_p, hit_g = (bp & wide[:num_obj]).flatten(1).sum(1), (bg & wide[num_obj:]).flatten(1).sum(1)
    n_p, n_g = bp.flatten(1).sum(1), bg.flatten(1).sum(1)
    rows = torch.stack([present.long(), inter, union, hit_p, hit_g, n_p, n_g], 1).tolist()      # one read-back
    js, fs = [], []
    for there, i, u, hp, hg, np_, ng in rows:
        if not there:
            continue
        js.append(i / u if u else 1.0)
        if np_ == 0 and ng == 0:
            fs.append(1.0)
            continue
        prec, rec = hp / max(1, np_), hg / max(1, ng)
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
