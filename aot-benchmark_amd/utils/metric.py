"""Region similarity J and boundary accuracy F between two label maps (the DAVIS measures the reference's README
defers to external toolkits for; `pytorch_iou`, reference utils/metric.py:4-36, is the only in-repo J-like
metric).  Used to report "J&F vs reference": masks of this engine scored against the reference's own masks."""
import torch
import torch.nn.functional as F


def _boundary(mask):
    """Boundary map of boolean [..., H, W] masks as the DAVIS toolkit's seg2bmap forms it (davis2017-evaluation, metrics.py): a pixel is
    a boundary pixel when its right, lower or lower-right neighbour differs; the last row / column compare with the one neighbour they
    have, the last pixel is never one."""
    e, s_, se = torch.zeros_like(mask), torch.zeros_like(mask), torch.zeros_like(mask)
    e[..., :, :-1] = mask[..., :, 1:]
    s_[..., :-1, :] = mask[..., 1:, :]
    se[..., :-1, :-1] = mask[..., 1:, 1:]
    b = (mask ^ e) | (mask ^ s_) | (mask ^ se)
    b[..., -1, :] = mask[..., -1, :] ^ e[..., -1, :]
    b[..., :, -1] = mask[..., :, -1] ^ s_[..., :, -1]
    b[..., -1, -1] = False
    return b


def _dilate(b, r):
    """Dilation by a disk of radius r (x^2 + y^2 <= r^2: skimage.morphology.disk, the toolkit's structuring element); b: boolean
    [..., H, W].  The disk is a stack of horizontal runs, one per row offset dy, of half width floor(sqrt(r^2 - dy^2)): each run is a
    1-D max filter of the map shifted by dy rows (ATen pooling kernels only -- no convolution library is involved)."""
    if r <= 0:
        return b
    H, W = b.shape[-2:]
    x = b.to(torch.float32).reshape(-1, 1, H, W)
    out = torch.zeros_like(x)
    for dy in range(-r, r + 1):
        hw = int((r * r - dy * dy) ** 0.5)
        while (hw + 1) ** 2 + dy * dy <= r * r:      # (exact integer half width whatever the float square root did)
            hw += 1
        while hw * hw + dy * dy > r * r:
            hw -= 1
        run = F.max_pool2d(x, (1, 2 * hw + 1), 1, (0, hw)) if hw > 0 else x
        # out[y] |= run[y - dy]: a source pixel at row y - dy reaches row y
        if dy >= 0:
            if dy < H:
                out[..., dy:, :] = torch.maximum(out[..., dy:, :], run[..., :H - dy, :])
        elif -dy < H:
            out[..., :H + dy, :] = torch.maximum(out[..., :H + dy, :], run[..., -dy:, :])
    return (out > 0.5).reshape(b.shape)


def jf_per_object(pred, ref, num_obj, bound_th=0.008):
    """pred, ref: integer label maps [H,W]; returns (J, F) averaged over objects 1..num_obj present in either map.
    F follows the DAVIS toolkit's f_measure (restated from its published code; the toolkit is not installed here: tests/test_host.py
    checks this implementation against an independent numpy / scipy restatement): seg2bmap boundaries, a tolerance of
    ceil(bound_th * ||(H, W)||) pixels as a disk dilation, precision / recall of the boundary pixels, their special cases for empty
    boundaries.  All objects are scored in one pass over [num_obj, H, W] stacks."""
    H, W = ref.shape[-2:]
    r = int(bound_th) if bound_th >= 1 else int(-(-(bound_th * (H * H + W * W) ** 0.5) // 1))
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
        if np_ == 0 and ng > 0:
            prec, rec = 1.0, 0.0
        elif np_ > 0 and ng == 0:
            prec, rec = 0.0, 1.0
        elif np_ == 0 and ng == 0:
            prec, rec = 1.0, 1.0
        else:
            prec, rec = hp / np_, hg / ng
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
