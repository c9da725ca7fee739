"""Small building blocks of the LSTT / decoder (reference networks/layers/basic.py).
The modules only own parameters (reference state_dict names); compute is in csrc/ via aot_hip."""
from torch import nn


class GroupNorm1D(nn.Module):  # basic.py:6-12
    def __init__(self, indim, groups=8):
        super().__init__()
        self.gn = nn.GroupNorm(groups, indim)


class GNActDWConv2d(nn.Module):  # basic.py:15-35: GroupNorm(32) -> exact GELU -> 5x5 depthwise conv (no bias)
    def __init__(self, indim, gn_groups=32):
        super().__init__()
        self.gn = nn.GroupNorm(gn_groups, indim)
        self.conv = nn.Conv2d(indim, indim, 5, dilation=1, padding=2, groups=indim, bias=False)


class DWConv2d(nn.Module):  # basic.py:38-57
    def __init__(self, indim, dropout=0.1):
        super().__init__()
        self.conv = nn.Conv2d(indim, indim, 5, dilation=1, padding=2, groups=indim, bias=False)
        self.dropout_p = dropout          # nn.Dropout2d after the conv: training-time only (models/train_forward.py)


class ConvGN(nn.Module):  # basic.py:75-85: conv (with bias) + GroupNorm(8)
    def __init__(self, indim, outdim, kernel_size, gn_groups=8):
        super().__init__()
        self.conv = nn.Conv2d(indim, outdim, kernel_size, padding=kernel_size // 2)
        self.gn = nn.GroupNorm(gn_groups, outdim)


def seq_to_2d(tensor, size_2d):
    """[N,B,C] -> [B,C,h,w] (basic.py:88-92).  Token-major memory IS channels-last memory, so this is a
    zero-copy view here (the reference makes a contiguous NCHW copy)."""
    h, w = size_2d
    _, n, c = tensor.size()
    return tensor.view(h, w, n, c).permute(2, 3, 0, 1)
