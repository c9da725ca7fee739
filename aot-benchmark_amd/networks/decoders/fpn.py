"""FPN segmentation head (reference networks/decoders/fpn.py:7-63).

HIP path (token-major maps): conv_in 1x1 GEMM -> GN(8)+ReLU -> adapter_16x GEMM with the previous map as
residual (epilogue add) -> 3x3 implicit GEMM -> GN+ReLU -> [bilinear up + adapter add fused] -> ... -> conv_out."""
import torch
from torch import nn

import aot_hip
from networks.layers.basic import ConvGN
from networks.layers.normalization import fold_conv_bn


class FPNSegmentationHead(nn.Module):
    def __init__(self, in_dim, out_dim, decode_intermediate_input=True, hidden_dim=256,
                 shortcut_dims=[24, 32, 96, 1280], align_corners=True):
        super().__init__()
        self.align_corners = align_corners
        self.decode_intermediate_input = decode_intermediate_input
        self.in_dim, self.out_dim, self.hidden = in_dim, out_dim, hidden_dim
        self.conv_in = ConvGN(in_dim, hidden_dim, 1)
        self.conv_16x = ConvGN(hidden_dim, hidden_dim, 3)
        self.conv_8x = ConvGN(hidden_dim, hidden_dim // 2, 3)
        self.conv_4x = ConvGN(hidden_dim // 2, hidden_dim // 2, 3)
        self.adapter_16x = nn.Conv2d(shortcut_dims[-2], hidden_dim, 1)
        self.adapter_8x = nn.Conv2d(shortcut_dims[-3], hidden_dim, 1)
        self.adapter_4x = nn.Conv2d(shortcut_dims[-4], hidden_dim // 2, 1)
        self.conv_out = nn.Conv2d(hidden_dim // 2, out_dim, 1)
        self._p = None

    def pack(self):
        if self._p is None:
            p = {}
            for n in ('conv_in', 'conv_16x', 'conv_8x', 'conv_4x'):
                m = getattr(self, n)
                p[n] = fold_conv_bn(m.conv)
                p[n + '_gn'] = (m.gn.weight.detach().float().contiguous(), m.gn.bias.detach().float().contiguous())
            for n in ('adapter_16x', 'adapter_8x', 'adapter_4x', 'conv_out'):
                p[n] = fold_conv_bn(getattr(self, n))
            self._p = p
        return self._p

    def _gn_relu(self, x, key, ws, stream):
        p = self._p
        dev = x.device
        aot_hip.groupnorm(x, *p[key + '_gn'], x, 8, ws.get('gn_scratch', (32 * 64 * 2,), dev, torch.float64),
                          ws.get('gn_stats', (64,), dev, torch.float64), act=aot_hip.ACT_RELU, nsplit=64, stream=stream)
        return x

    def run(self, x_in, f16, f8, f4, ws, stream):
        """x_in [N16, in_dim] (concatenated decoder input, or the last LSTT output), f16/f8/f4 = (feat, h, w)
        shortcuts at strides 16/8/4.  Returns logits [h4*w4, out_dim] (row stride out_dim padded to 4)."""
        p = self.pack()
        dev = x_in.device
        hd = self.hidden
        (s16, h16, w16), (s8, h8, w8), (s4, h4, w4) = f16, f8, f4
        n16, n8, n4 = h16 * w16, h8 * w8, h4 * w4
        a = ws.get('dec_a16', (n16, hd), dev)
        aot_hip.conv2d(x_in, *p['conv_in'], a, h16, w16, x_in.shape[1], h16, w16, hd, stream=stream)
        self._gn_relu(a, 'conv_in', ws, stream)
        b = ws.get('dec_b16', (n16, hd), dev)
        aot_hip.conv2d(s16, *p['adapter_16x'], b, h16, w16, s16.shape[1], h16, w16, hd, res=a, stream=stream)
        aot_hip.conv2d(b, *p['conv_16x'], a, h16, w16, hd, h16, w16, hd, 3, 3, 1, 1, 1, stream=stream)
        self._gn_relu(a, 'conv_16x', ws, stream)
        # 8x: adapter(shortcut) + bilinear(a)
        c = ws.get('dec_a8', (n8, hd), dev)
        aot_hip.conv2d(s8, *p['adapter_8x'], c, h8, w8, s8.shape[1], h8, w8, hd, stream=stream)
        aot_hip.bilinear(a, c, h16, w16, h8, w8, hd, self.align_corners, add=c, stream=stream)
        d = ws.get('dec_b8', (n8, hd // 2), dev)
        aot_hip.conv2d(c, *p['conv_8x'], d, h8, w8, hd, h8, w8, hd // 2, 3, 3, 1, 1, 1, stream=stream)
        self._gn_relu(d, 'conv_8x', ws, stream)
        # 4x
        e = ws.get('dec_a4', (n4, hd // 2), dev)
        aot_hip.conv2d(s4, *p['adapter_4x'], e, h4, w4, s4.shape[1], h4, w4, hd // 2, stream=stream)
        aot_hip.bilinear(d, e, h8, w8, h4, w4, hd // 2, self.align_corners, add=e, stream=stream)
        f = ws.get('dec_b4', (n4, hd // 2), dev)
        aot_hip.conv2d(e, *p['conv_4x'], f, h4, w4, hd // 2, h4, w4, hd // 2, 3, 3, 1, 1, 1, stream=stream)
        self._gn_relu(f, 'conv_4x', ws, stream)
        ldo = (self.out_dim + 3) // 4 * 4
        out = ws.get('dec_logits', (n4, ldo), dev)
        aot_hip.conv2d(f, *p['conv_out'], out, h4, w4, hd // 2, h4, w4, self.out_dim, stream=stream)
        return out[:, :self.out_dim], h4, w4
