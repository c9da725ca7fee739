"""MobileNetV2 at output stride 16 (reference networks/encoders/mobilenetv2.py:116-247).

HIP path: pointwise convs are fp32-MFMA GEMMs with folded BN and a fused ReLU6 / residual epilogue, the
3x3 depthwise convs (stride 1/2, dilation 1/2) run in the NHWC depthwise kernel with folded BN + ReLU6."""
from torch import nn

import aot_hip
from networks.layers.normalization import fold_conv_bn, fold_dwconv_bn


def _osz(n, k, s, p, d=1):
    return (n + 2 * p - d * (k - 1) - 1) // s + 1


class ConvBNActivation(nn.Sequential):
    def __init__(self, in_planes, out_planes, kernel_size=3, stride=1, groups=1, norm_layer=None, dilation=1):
        padding = (kernel_size - 1) // 2 * dilation
        super().__init__(nn.Conv2d(in_planes, out_planes, kernel_size, stride, padding, dilation=dilation,
                                   groups=groups, bias=False), norm_layer(out_planes), nn.ReLU6(inplace=True))
        self.out_channels = out_planes


ConvBNReLU = ConvBNActivation


class InvertedResidual(nn.Module):
    def __init__(self, inp, oup, stride, dilation, expand_ratio, norm_layer=None):
        super().__init__()
        self.stride = stride
        self.dilation = dilation
        hidden_dim = int(round(inp * expand_ratio))
        self.use_res_connect = self.stride == 1 and inp == oup
        layers = []
        if expand_ratio != 1:
            layers.append(ConvBNReLU(inp, hidden_dim, kernel_size=1, norm_layer=norm_layer))
        layers.extend([ConvBNReLU(hidden_dim, hidden_dim, stride=stride, dilation=dilation, groups=hidden_dim,
                                  norm_layer=norm_layer),
                       nn.Conv2d(hidden_dim, oup, 1, 1, 0, bias=False), norm_layer(oup)])
        self.conv = nn.Sequential(*layers)
        self.expand = expand_ratio != 1
        self.hidden, self.inp, self.oup = hidden_dim, inp, oup
        self._p = None

    def pack(self):
        if self._p is None:
            j = 1 if self.expand else 0
            p = {'dw': fold_dwconv_bn(self.conv[j][0], self.conv[j][1]),
                 'pl': fold_conv_bn(self.conv[j + 1], self.conv[j + 2])}
            if self.expand:
                p['pw'] = fold_conv_bn(self.conv[0][0], self.conv[0][1])
            self._p = p
        return self._p

    def run(self, x, H, W, ws, stream, tag):
        p = self.pack()
        dev = x.device
        y = x
        if self.expand:
            y = ws.get('ir_pw', (H * W, self.hidden), dev)
            aot_hip.conv2d(x, *p['pw'], y, H, W, self.inp, H, W, self.hidden, act=aot_hip.ACT_RELU6, stream=stream)
        s, d = self.stride, self.dilation
        OH, OW = _osz(H, 3, s, d, d), _osz(W, 3, s, d, d)
        z = ws.get('ir_dw', (OH * OW, self.hidden), dev)
        aot_hip.dwconv2d(y, p['dw'][0], p['dw'][1], z, H, W, self.hidden, OH, OW, 3, s, d, d, act=aot_hip.ACT_RELU6,
                         stream=stream)
        out = ws.get('ir_out' + tag, (OH * OW, self.oup), dev)
        aot_hip.conv2d(z, *p['pl'], out, OH, OW, self.hidden, OH, OW, self.oup,
                       res=x if self.use_res_connect else None, stream=stream)
        return out, OH, OW


class MobileNetV2(nn.Module):
    def __init__(self, output_stride=8, norm_layer=None, width_mult=1.0, inverted_residual_setting=None,
                 round_nearest=8, block=None, freeze_at=0):
        super().__init__()
        if width_mult != 1.0 or inverted_residual_setting is not None:
            raise NotImplementedError
        setting = [[1, 16, 1, 1], [6, 24, 2, 2], [6, 32, 3, 2], [6, 64, 4, 2], [6, 96, 3, 1], [6, 160, 3, 2],
                   [6, 320, 1, 1]]
        input_channel, self.last_channel = 32, 1280
        features = [ConvBNReLU(3, input_channel, stride=2, norm_layer=norm_layer)]
        current_stride, rate = 2, 1
        for t, c, n, s in setting:          # stride/dilation bookkeeping of mobilenetv2.py:180-201
            if current_stride == output_stride:
                stride, dilation = 1, rate
                rate *= s
            else:
                stride, dilation = s, 1
                current_stride *= s
            for i in range(n):
                if i == 0:
                    features.append(InvertedResidual(input_channel, c, stride, dilation, t, norm_layer))
                else:
                    features.append(InvertedResidual(input_channel, c, 1, rate, t, norm_layer))
                input_channel = c
        features.append(ConvBNReLU(input_channel, self.last_channel, kernel_size=1, norm_layer=norm_layer))
        self.features = nn.Sequential(*features)
        self._stem = self._last = None
        self.freeze(freeze_at)

    def freeze(self, freeze_at):
        """reference mobilenetv2.py:240-247: stages = features[0:4], [4:7], [7:14], [14:]; freeze_at >= 1 stops the gradients of
        the first conv block, freeze_at >= k+1 those of stage k."""
        stages = [self.features[0:4], self.features[4:7], self.features[7:14], self.features[14:]]
        frozen = ([self.features[0]] if freeze_at >= 1 else []) + [st for idx, st in enumerate(stages, start=2) if freeze_at >= idx]
        for m in frozen:
            for p in m.parameters():
                p.requires_grad = False

    def run(self, img, ws, stream):
        """img [1,3,H,W] -> [(feat, h, w)] for the 4 stages features[0:4],[4:7],[7:14],[14:] (:210-224)."""
        _, _, H, W = img.shape
        dev = img.device
        if self._stem is None:
            self._stem = fold_conv_bn(self.features[0][0], self.features[0][1], pad_cin=4)
            self._last = fold_conv_bn(self.features[18][0], self.features[18][1])
        x4 = ws.get('img_nhwc4', (H * W, 4), dev)
        aot_hip.nchw_to_nhwc(img, x4, 3, H, W, 4, stream=stream)
        h, w = _osz(H, 3, 2, 1), _osz(W, 3, 2, 1)
        x = ws.get('mb_stem', (h * w, 32), dev)
        aot_hip.conv2d(x4, *self._stem, x, H, W, 4, h, w, 32, 3, 3, 2, 1, 1, act=aot_hip.ACT_RELU6, stream=stream)
        feats = []
        for idx in range(1, 18):
            stage_end = idx in (3, 6, 13)
            x, h, w = self.features[idx].run(x, h, w, ws, stream, '_s%d' % idx if stage_end else '_%d' % (idx & 1))
            if stage_end:
                feats.append((x, h, w))
        y = ws.get('mb_last', (h * w, self.last_channel), dev)
        aot_hip.conv2d(x, *self._last, y, h, w, x.shape[1], h, w, self.last_channel, act=aot_hip.ACT_RELU6, stream=stream)
        feats.append((y, h, w))
        return feats
