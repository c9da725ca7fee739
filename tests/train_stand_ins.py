"""Test infrastructure: plain-torch stand-ins for the differentiable primitives of networks/layers/train_ops.py (same
signatures, same token-major layouts), so that the GLUE of the training graph (networks/models/train_forward.py: which op
feeds which, weight layouts, the frame recurrence) can be held to the reference's gradient goldens on CPU, where the HIP
kernels cannot run.  The kernels themselves are checked one by one on the GPU (tests/test_training_gpu.py).  Nothing under
aot-benchmark_amd/ imports this file."""
import torch
import torch.nn.functional as F


def matmul(a, b, bias=None, alpha=1.0):
    c = torch.matmul(a, b) * alpha
    return c if bias is None else c + bias


def linear(x, weight, bias=None):
    return F.linear(x, weight, bias)


def _maps(x, B, H, W):
    return x.view(B, H, W, x.shape[1]).permute(0, 3, 1, 2)


def _tokens(y):
    B, C, H, W = y.shape
    return y.permute(0, 2, 3, 1).reshape(B * H * W, C)


def conv2d(x, weight, bias, B, H, W, stride=1, pad=0, dil=1):
    cin = weight.shape[1]
    if x.shape[1] != cin:
        weight = F.pad(weight, (0, 0, 0, 0, 0, x.shape[1] - cin))
    y = F.conv2d(_maps(x, B, H, W), weight, bias, stride, pad, dil)
    return _tokens(y), y.shape[2], y.shape[3]


def dwconv2d(x, weight, B, H, W, stride=1, pad=0, dil=1):
    y = F.conv2d(_maps(x, B, H, W), weight, None, stride, pad, dil, groups=weight.shape[0])
    return _tokens(y), y.shape[2], y.shape[3]


def maxpool3x3s2(x, H, W):
    y = F.max_pool2d(_maps(x, 1, H, W), 3, 2, 1)
    return _tokens(y), y.shape[2], y.shape[3]


def act(x, kind):
    return {'none': lambda t: t, 'relu': F.relu, 'relu6': F.relu6, 'gelu': F.gelu, 'silu': F.silu}[kind](x)


def layernorm(x, gamma, beta, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), gamma, beta, eps)


def groupnorm(x, gamma, beta, groups, B=1, eps=1e-5):
    R, C = x.shape
    y = F.group_norm(x.view(B, R // B, C).permute(0, 2, 1), groups, gamma, beta, eps)
    return y.permute(0, 2, 1).reshape(R, C)


def softmax_rows(x):
    return torch.softmax(x, -1)


def bilinear(x, B, IH, IW, OH, OW, align_corners):
    return _tokens(F.interpolate(_maps(x, B, IH, IW), size=(OH, OW), mode='bilinear', align_corners=bool(align_corners)))


def _window_index(h, w, R, device):
    """key index [N, W2] of every query's window entry (dy, dx) -> (y + dy - R, x + dx - R), and its in-image mask."""
    ws = 2 * R + 1
    ys = torch.arange(h, device=device).view(h, 1, 1, 1)
    xs = torch.arange(w, device=device).view(1, w, 1, 1)
    ky = ys + torch.arange(ws, device=device).view(1, 1, ws, 1) - R
    kx = xs + torch.arange(ws, device=device).view(1, 1, 1, ws) - R
    valid = ((ky >= 0) & (ky < h) & (kx >= 0) & (kx < w)).reshape(h * w, ws * ws)
    idx = (ky.clamp(0, h - 1) * w + kx.clamp(0, w - 1)).reshape(h * w, ws * ws)
    return idx, valid


def window_gather(dense, h, w, max_dis, fill=0.0):
    idx, valid = _window_index(h, w, max_dis, dense.device)
    G = dense.shape[0]
    got = torch.gather(dense, 2, idx.unsqueeze(0).expand(G, -1, -1))
    return torch.where(valid.unsqueeze(0), got, torch.full_like(got, fill))


def window_scatter(win, h, w, max_dis, fill=0.0):
    idx, valid = _window_index(h, w, max_dis, win.device)
    G, N = win.shape[0], h * w
    idx = torch.where(valid, idx, torch.full_like(idx, N))                     # entries outside the image go to a spare column
    dense = win.new_full((G, N, N + 1), fill)
    dense = dense.scatter(2, idx.unsqueeze(0).expand(G, -1, -1), win)
    return dense[:, :, :N]


def to_nchw(x, H, W):
    return x.view(H, W, x.shape[1]).permute(2, 0, 1).unsqueeze(0)


def to_nhwc(x, cpad=None):
    _, C, H, W = x.shape
    y = x[0].permute(1, 2, 0).reshape(H * W, C)
    return y if cpad is None or cpad == C else F.pad(y, (0, cpad - C))


def permute_cols(x, perm):
    return x[:, perm.to(x.device).long()]


def install(monkeypatch):
    """Replaces the primitives of networks.layers.train_ops by the stand-ins above for one test."""
    from networks.layers import train_ops
    for name in ('matmul', 'linear', 'conv2d', 'dwconv2d', 'act', 'layernorm', 'groupnorm', 'softmax_rows', 'bilinear',
                 'window_gather', 'window_scatter', 'to_nchw', 'to_nhwc', 'maxpool3x3s2', 'permute_cols'):
        monkeypatch.setattr(train_ops, name, globals()[name])
