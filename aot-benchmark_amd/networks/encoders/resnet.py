"""ResNet-50/101 trunk without layer4, output stride 16 (reference networks/encoders/resnet.py:57-175).

HIP path: NHWC activations; every conv+FrozenBN(+ReLU)(+residual) is ONE launch of the fp32-MFMA
implicit-GEMM kernel (BN folded into weights/bias at pack time, residual add and ReLU in the epilogue);
the stem is a 7x7/s2 implicit GEMM on a 4-channel-padded image followed by a 3x3/s2 max pool.
"""
import torch
from torch import nn

import aot_hip
from networks.layers.normalization import fold_conv_bn


def _osz(n, k, s, p, d=1):
    return (n + 2 * p - d * (k - 1) - 1) // s + 1


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, BatchNorm=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = BatchNorm(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, dilation=dilation, padding=dilation,
                               bias=False)
        self.bn2 = BatchNorm(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = BatchNorm(planes * 4)
        self.downsample = downsample
        self.stride = stride
        self.dilation = dilation
        self._p = None

    def pack(self):
        if self._p is None:
            p = {'c1': fold_conv_bn(self.conv1, self.bn1), 'c2': fold_conv_bn(self.conv2, self.bn2),
                 'c3': fold_conv_bn(self.conv3, self.bn3)}
            if self.downsample is not None:
                p['ds'] = fold_conv_bn(self.downsample[0], self.downsample[1])
            self._p = p
        return self._p

    def run(self, x, H, W, ws, stream, out=None, tag='', B=1):
        """x [B*H*W, Cin] (B images stacked along the rows) -> [B*OH*OW, 4*planes]; reference Bottleneck.forward,
        resnet.py:34-54."""
        p = self.pack()
        dev = x.device
        cin = x.shape[1]
        planes = self.conv1.out_channels
        s, d = self.stride, self.dilation
        OH, OW = _osz(H, 3, s, d, d), _osz(W, 3, s, d, d)
        t1 = ws.get('bt1' + tag, (B * H * W, planes), dev)
        aot_hip.conv2d(x, *p['c1'], t1, H, W, cin, H, W, planes, act=aot_hip.ACT_RELU, B=B, stream=stream)
        t2 = ws.get('bt2' + tag, (B * OH * OW, planes), dev)
        aot_hip.conv2d(t1, *p['c2'], t2, H, W, planes, OH, OW, planes, 3, 3, s, d, d, act=aot_hip.ACT_RELU, B=B,
                       stream=stream)
        if self.downsample is not None:
            res = ws.get('bds' + tag, (B * OH * OW, planes * 4), dev)
            aot_hip.conv2d(x, *p['ds'], res, H, W, cin, OH, OW, planes * 4, 1, 1, s, 0, 1, B=B, stream=stream)
        else:
            res = x
        if out is None:
            out = ws.get('bout' + tag, (B * OH * OW, planes * 4), dev)
        aot_hip.conv2d(t2, *p['c3'], out, OH, OW, planes, OH, OW, planes * 4, res=res, act=aot_hip.ACT_RELU, B=B,
                       stream=stream)
        return out, OH, OW


class ResNet(nn.Module):
    def __init__(self, block, layers, output_stride, BatchNorm, freeze_at=0):
        super().__init__()
        if output_stride != 16:
            raise NotImplementedError
        self.inplanes = 64
        strides, dilations = [1, 2, 2, 1], [1, 1, 1, 2]
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = BatchNorm(64)
        self.layer1 = self._make_layer(block, 64, layers[0], strides[0], dilations[0], BatchNorm)
        self.layer2 = self._make_layer(block, 128, layers[1], strides[1], dilations[1], BatchNorm)
        self.layer3 = self._make_layer(block, 256, layers[2], strides[2], dilations[2], BatchNorm)
        self._stem = None
        self.freeze(freeze_at)

    def freeze(self, freeze_at):
        """Stops gradients of the stem (freeze_at >= 1) and of layer{1,2,3} (freeze_at >= 2, 3, 4): reference resnet.py:168-175
        (TRAIN_ENCODER_FREEZE_AT; decides which tensors the trainer's parameter groups hold)."""
        frozen = ([self.conv1, self.bn1] if freeze_at >= 1 else []) + \
                 [st for idx, st in enumerate((self.layer1, self.layer2, self.layer3), start=2) if freeze_at >= idx]
        for m in frozen:
            for p in m.parameters():
                p.requires_grad = False

    def _make_layer(self, block, planes, blocks, stride, dilation, BatchNorm):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                BatchNorm(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, max(dilation // 2, 1), downsample, BatchNorm)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, dilation=dilation, BatchNorm=BatchNorm))
        return nn.Sequential(*layers)

    batched = True       # run() takes B images at once (AOTEngine.encode_ahead: the frames ahead of a clip as one batch)

    def run(self, img, ws, stream):
        """img [B,3,H,W] planar -> [(feat [B*h*w, C], h, w)] for strides 4, 8, 16, the B images stacked along the rows
        (reference forward, :140-157; the reference returns the stride-16 map twice).  The reference's offline_encoder
        (aot_engine.py:147-166) likewise encodes all frames it has in one batch."""
        B, _, H, W = img.shape
        dev = img.device
        if self._stem is None:
            self._stem = fold_conv_bn(self.conv1, self.bn1, pad_cin=4)
            self._stem[0]._aot_c4_taps = 49          # (bf16x6 engines: its four-channel planes are packed by pack_bf16x6_all; fold_conv_bn registered it)
        x4 = ws.get('img_nhwc4', (B * H * W, 4), dev)
        H1, W1 = _osz(H, 7, 2, 3), _osz(W, 7, 2, 3)
        for b in range(B):
            aot_hip.nchw_to_nhwc(img[b:b + 1], x4[b * H * W:(b + 1) * H * W], 3, H, W, 4, stream=stream)
        s1 = ws.get('stem', (B * H1 * W1, 64), dev)
        aot_hip.conv2d_c4(x4, *self._stem, s1, H, W, H1, W1, 64, 7, 7, 2, 3, 1, act=aot_hip.ACT_RELU, B=B, stream=stream)
        H2, W2 = _osz(H1, 3, 2, 1), _osz(W1, 3, 2, 1)
        x = ws.get('pool', (B * H2 * W2, 64), dev)
        for b in range(B):
            aot_hip.maxpool3x3s2(s1[b * H1 * W1:(b + 1) * H1 * W1], x[b * H2 * W2:(b + 1) * H2 * W2], H1, W1, 64, H2, W2,
                                 stream=stream)
        feats = []
        h, w = H2, W2
        for li, layer in enumerate((self.layer1, self.layer2, self.layer3)):
            nb = len(layer)
            for bi, blk in enumerate(layer):
                last = bi == nb - 1
                out = None
                if last:   # stage outputs are decoder shortcuts: keep them in their own buffers
                    planes4 = blk.conv3.out_channels
                    ho, wo = _osz(h, 3, blk.stride, blk.dilation, blk.dilation), _osz(w, 3, blk.stride, blk.dilation, blk.dilation)
                    out = ws.get('stage%d' % li, (B * ho * wo, planes4), dev)
                # ping-pong block outputs so a block never overwrites its own input / residual
                x, h, w = blk.run(x, h, w, ws, stream, out=out, tag='_%d_%d' % (li, bi & 1), B=B)
            feats.append((x, h, w))
        return feats


def ResNet50(output_stride, BatchNorm, freeze_at=0):
    return ResNet(Bottleneck, [3, 4, 6, 3], output_stride, BatchNorm, freeze_at=freeze_at)


def ResNet101(output_stride, BatchNorm, freeze_at=0):
    return ResNet(Bottleneck, [3, 4, 23, 3], output_stride, BatchNorm, freeze_at=freeze_at)
