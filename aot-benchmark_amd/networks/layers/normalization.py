"""FrozenBatchNorm2d parameter holder (reference networks/layers/normalization.py:6-17).

On the HIP path the affine transform is folded into the preceding convolution's weights and
bias once, at ``prepare()`` time, so no kernel is ever launched for it."""
import torch
from torch import nn

from aot_hip import attach_wt


class FrozenBatchNorm2d(nn.Module):
    def __init__(self, n, epsilon=1e-5):
        super().__init__()
        self.register_buffer('weight', torch.ones(n))
        self.register_buffer('bias', torch.zeros(n))
        self.register_buffer('running_mean', torch.zeros(n))
        self.register_buffer('running_var', torch.ones(n) - epsilon)
        self.epsilon = epsilon

    def fold(self):
        """(scale, shift) in float64 such that bn(x) = x*scale + shift."""
        scale = self.weight.double() * (self.running_var.double() + self.epsilon).rsqrt()
        return scale, self.bias.double() - self.running_mean.double() * scale


def fold_conv_bn(conv, bn=None, pad_cin=None):
    """Conv2d (+ FrozenBN) -> (W [KH*KW*Cin, Cout_pad4] fp32, bias [Cout] fp32) for the implicit-GEMM kernel.
    k index = (ky*KW + kx)*Cin + c, matching the NHWC im2col order of csrc/gemm_conv.hip."""
    w = conv.weight.detach().double()                       # [Cout, Cin, KH, KW]
    cout, cin, kh, kw = w.shape
    b = conv.bias.detach().double() if conv.bias is not None else torch.zeros(cout, dtype=torch.float64, device=w.device)
    if bn is not None:
        scale, shift = bn.fold()
        w = w * scale.view(-1, 1, 1, 1)
        b = b * scale + shift
    if pad_cin is not None and pad_cin > cin:
        w = torch.cat([w, w.new_zeros(cout, pad_cin - cin, kh, kw)], 1)
        cin = pad_cin
    wk = w.permute(2, 3, 1, 0).reshape(kh * kw * cin, cout)
    ldb = (cout + 3) // 4 * 4
    out = torch.zeros(kh * kw * cin, ldb, dtype=torch.float32, device=w.device)
    out[:, :cout] = wk.float()
    return attach_wt(out, cin), b.float().contiguous()


def fold_dwconv_bn(conv, bn=None):
    """Depthwise Conv2d [C,1,K,K] (+ FrozenBN) -> (W [K*K, C], bias [C] or None)."""
    w = conv.weight.detach().double()
    c, _, kh, kw = w.shape
    b = conv.bias.detach().double() if conv.bias is not None else None
    if bn is not None:
        scale, shift = bn.fold()
        w = w * scale.view(-1, 1, 1, 1)
        b = shift if b is None else b * scale + shift
    wk = w.view(c, kh * kw).t().contiguous().float()
    return wk, (None if b is None else b.float().contiguous())


def linear_t(lin):
    """nn.Linear -> (W^T [in, out] fp32 contiguous, bias)."""
    return (attach_wt(lin.weight.detach().t().contiguous().float()),
            (None if lin.bias is None else lin.bias.detach().float().contiguous()))
