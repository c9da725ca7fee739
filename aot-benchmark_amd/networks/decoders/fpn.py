"""FPN segmentation head (reference networks/decoders/fpn.py:7-63).

HIP path (token-major maps): conv_in 1x1 GEMM -> GN(8)+ReLU -> adapter_16x GEMM with the previous map as
residual (epilogue add) -> 3x3 implicit GEMM -> GN+ReLU -> [bilinear up + adapter add fused] -> ... -> conv_out."""
import os

import torch
from torch import nn

import aot_hip
from networks.layers.basic import ConvGN
from networks.layers.normalization import fold_conv_bn


# row ranges of a GroupNorm statistics launch: >= 64 selects the whole-row kernel (gn_stats_rows_kernel, round 6: +0.6 % over the per-group kernel
# at 32 ranges, profiles/r06_gn_stats_rows.txt)
GN_SPLIT = 128


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

    def _gn_relu(self, x, out, key, B, ws, stream, add=None, add_rows=0):
        """relu(gn(x)) (+ add): the statistics launch, then the apply launch."""
        p = self._p
        aot_hip.groupnorm(x, *p[key + '_gn'], out, 8, aot_hip.gn_buffers(ws, x.device, B, 8, GN_SPLIT), act=aot_hip.ACT_RELU,
                          eps=getattr(self, key).gn.eps, nsplit=GN_SPLIT, B=B, add=add, add_rows=add_rows, stream=stream)
        return out

    def _gn_relu_up(self, x, out, key, B, ws, stream, ih, iw, oh, ow, add):
        """bilinear(relu(gn(x))) + add: the statistics launch, then ONE launch that normalises the four taps of the resize on the fly
        (aot_gn_bilinear_nhwc_f32, bit-identical to apply + resize): the normalised map of a block whose only consumer is the next
        upsampling is never written.  AOT_NO_GN_UP: the separate apply + resize launches (A/B runs)."""
        p = self._p
        if os.environ.get('AOT_NO_GN_UP'):
            self._gn_relu(x, x, key, B, ws, stream)
            return aot_hip.bilinear(x, out, ih, iw, oh, ow, x.shape[1], self.align_corners, add=add, B=B, add_shared=True, stream=stream)
        stats = aot_hip.groupnorm_stats(x, 8, aot_hip.gn_buffers(ws, x.device, B, 8, GN_SPLIT), B=B, eps=getattr(self, key).gn.eps, nsplit=GN_SPLIT,
                                        stream=stream)
        return aot_hip.gn_bilinear(x, stats, *p[key + '_gn'], out, ih, iw, oh, ow, x.shape[1], 8, self.align_corners,
                                   act=aot_hip.ACT_RELU, add=add, B=B, add_shared=True, stream=stream)

    def adapters(self, f16, f8, f4, ws, stream, B=1):
        """The three adapter convolutions (fpn.py:36-37,45-46,52-53) of B stacked frames: they read the encoder's shortcut maps only,
        so the engine runs them WITH the encoder (batched over the look-ahead frames, round 5) instead of once per frame inside the
        decode stage.  Returns (ad16 [B*n16, hd], ad8 [B*n8, hd], ad4 [B*n4, hd/2]) in per-stream scratch."""
        p = self.pack()
        hd = self.hidden
        (s16, h16, w16), (s8, h8, w8), (s4, h4, w4) = f16, f8, f4
        dev = s16.device
        ad16 = ws.get('enc_ad16', (B * h16 * w16, hd), dev)
        aot_hip.conv2d(s16, *p['adapter_16x'], ad16, h16, w16, s16.shape[1], h16, w16, hd, B=B, stream=stream)
        ad8 = ws.get('enc_ad8', (B * h8 * w8, hd), dev)
        aot_hip.conv2d(s8, *p['adapter_8x'], ad8, h8, w8, s8.shape[1], h8, w8, hd, B=B, stream=stream)
        ad4 = ws.get('enc_ad4', (B * h4 * w4, hd // 2), dev)
        aot_hip.conv2d(s4, *p['adapter_4x'], ad4, h4, w4, s4.shape[1], h4, w4, hd // 2, B=B, stream=stream)
        return ad16, ad8, ad4

    def run(self, x_in, f16, f8, f4, ws, stream, B=1, ads=None):
        """x_in [B*N16, in_dim] (concatenated decoder input, or the last LSTT output) of B lanes (object groups of ONE
        frame), f16/f8/f4 = (feat, h, w) shortcuts at strides 16/8/4 shared by the lanes: the three adapter convs run
        once and are added to every lane (GN-apply epilogue at 16x, bilinear epilogue at 8x / 4x); ads = (ad16, ad8, ad4) of THIS
        frame when adapters() has already formed them.  Returns logits [B*h4*w4, out_dim] (row stride out_dim padded to 4)."""
        p = self.pack()
        dev = x_in.device
        hd = self.hidden
        if x_in.shape[1] != self.in_dim:      # conv_in's weight has in_dim rows: a wider input would read past it
            raise aot_hip.AotHipError('decoder expects %d input channels, got %d' % (self.in_dim, x_in.shape[1]))
        (s16, h16, w16), (s8, h8, w8), (s4, h4, w4) = f16, f8, f4
        n16, n8, n4 = h16 * w16, h8 * w8, h4 * w4
        if ads is not None:
            ad16 = ads[0]
        else:
            ad16 = ws.get('dec_ad16', (n16, hd), dev)
            aot_hip.conv2d(s16, *p['adapter_16x'], ad16, h16, w16, s16.shape[1], h16, w16, hd, stream=stream)
        a = ws.get('dec_a16', (B * n16, hd), dev)
        aot_hip.linear(x_in, *p['conv_in'], a, stream=stream)
        b = ws.get('dec_b16', (B * n16, hd), dev)
        self._gn_relu(a, b, 'conv_in', B, ws, stream, add=ad16, add_rows=n16)      # relu(gn(conv_in)) + adapter_16x
        aot_hip.conv2d(b, *p['conv_16x'], a, h16, w16, hd, h16, w16, hd, 3, 3, 1, 1, 1, B=B, stream=stream)
        # 8x: adapter(shortcut) + bilinear(a)
        if ads is not None:
            ad8 = ads[1]
        else:
            ad8 = ws.get('dec_ad8', (n8, hd), dev)
            aot_hip.conv2d(s8, *p['adapter_8x'], ad8, h8, w8, s8.shape[1], h8, w8, hd, stream=stream)
        c = ws.get('dec_a8', (B * n8, hd), dev)
        self._gn_relu_up(a, c, 'conv_16x', B, ws, stream, h16, w16, h8, w8, ad8)      # up(relu(gn(conv_16x))) + adapter_8x
        d = ws.get('dec_b8', (B * n8, hd // 2), dev)
        aot_hip.conv2d(c, *p['conv_8x'], d, h8, w8, hd, h8, w8, hd // 2, 3, 3, 1, 1, 1, B=B, stream=stream)
        # 4x
        if ads is not None:
            ad4 = ads[2]
        else:
            ad4 = ws.get('dec_ad4', (n4, hd // 2), dev)
            aot_hip.conv2d(s4, *p['adapter_4x'], ad4, h4, w4, s4.shape[1], h4, w4, hd // 2, stream=stream)
        e = ws.get('dec_a4', (B * n4, hd // 2), dev)
        self._gn_relu_up(d, e, 'conv_8x', B, ws, stream, h8, w8, h4, w4, ad4)
        f = ws.get('dec_b4', (B * n4, hd // 2), dev)
        aot_hip.conv2d(e, *p['conv_4x'], f, h4, w4, hd // 2, h4, w4, hd // 2, 3, 3, 1, 1, 1, B=B, stream=stream)
        ldo = (self.out_dim + 3) // 4 * 4
        out = ws.get('dec_logits', (B * n4, ldo), dev)
        if self.out_dim <= 32 and not os.environ.get('AOT_NO_GN_UP'):
            # conv_out reads relu(gn(conv_4x)) through its A loads (aot_gn_conv1x1_f32): the normalised 4x map is never written
            st = aot_hip.groupnorm_stats(f, 8, aot_hip.gn_buffers(ws, dev, B, 8, GN_SPLIT), B=B, eps=self.conv_4x.gn.eps, nsplit=GN_SPLIT, stream=stream)
            aot_hip.gn_conv1x1(f, st, *p['conv_4x_gn'], *p['conv_out'], out, 8, self.out_dim, gn_act=aot_hip.ACT_RELU, B=B, stream=stream)
        else:
            self._gn_relu(f, f, 'conv_4x', B, ws, stream)
            aot_hip.conv2d(f, *p['conv_out'], out, 1, B * n4, hd // 2, 1, B * n4, self.out_dim, stream=stream)
        return out[:, :self.out_dim], h4, w4
