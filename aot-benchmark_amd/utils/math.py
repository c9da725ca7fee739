"""Random helpers of the training side (reference utils/math.py)."""
import torch


def generate_permute_matrix(dim, num, keep_first=True, device=None, gpu_id=None):
    """`num` random dim x dim permutation matrices, [num, dim, dim] (utils/math.py:3-25): row o holds a single 1 in the column
    identity o is moved to; keep_first leaves index 0 (the background) in place.  The training engine keeps the same shuffle
    as index vectors (AOTEngine.id_shuffle[b][o] = column of row o); `permutation_to_matrix` converts."""
    if device is None:
        device = torch.device('cuda', gpu_id) if (gpu_id is not None and torch.cuda.is_available()) else \
            torch.device('cuda' if torch.cuda.is_available() else 'cpu')
    out = []
    for _ in range(num):
        if keep_first:
            perm = torch.cat([torch.zeros(1, dtype=torch.long, device=device), 1 + torch.randperm(dim - 1, device=device)])
        else:
            perm = torch.randperm(dim, device=device)
        out.append(torch.eye(dim, device=device)[perm])
    return torch.stack(out, 0)


def permutation_to_matrix(perm):
    """perm[o] = new index of identity o  ->  the [dim, dim] matrix M with M[o, perm[o]] = 1 ('bohw,bot->bthw' moves channel o
    to channel perm[o])."""
    m = torch.zeros(perm.numel(), perm.numel(), device=perm.device)
    m[torch.arange(perm.numel(), device=perm.device), perm] = 1.
    return m


def truncated_normal_(tensor, mean=0., std=0.02):
    """In place: standard normal truncated to (-2, 2) -- the first of four draws per element that lands inside, scaled by std and
    shifted by mean (utils/math.py:28-37)."""
    with torch.no_grad():
        draws = torch.randn(tuple(tensor.shape) + (4,), device=tensor.device)
        inside = (draws.abs() < 2)
        first = inside.to(torch.uint8).argmax(-1, keepdim=True)
        tensor.copy_(draws.gather(-1, first).squeeze(-1).mul_(std).add_(mean))
    return tensor
