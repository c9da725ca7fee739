"""Encoder factory (reference networks/encoders/__init__.py:10-35).  Built: resnet50 / resnet101,
mobilenetv2 and swin_base (BASELINE configs 1-4); the other backbones of the reference are not on the scoped path."""
from networks.encoders.mobilenetv2 import MobileNetV2
from networks.encoders.resnet import ResNet50, ResNet101
from networks.encoders.swin import build_swin_model
from networks.layers.normalization import FrozenBatchNorm2d


def build_encoder(name, frozen_bn=True, freeze_at=-1):
    if not frozen_bn:
        raise NotImplementedError('the inference path folds FrozenBatchNorm2d into the convolutions')
    if name == 'mobilenetv2':
        return MobileNetV2(16, FrozenBatchNorm2d, freeze_at=freeze_at)
    if name == 'resnet50':
        return ResNet50(16, FrozenBatchNorm2d, freeze_at=freeze_at)
    if name == 'resnet101':
        return ResNet101(16, FrozenBatchNorm2d, freeze_at=freeze_at)
    if 'swin' in name:
        return build_swin_model(name, freeze_at=freeze_at)
    raise NotImplementedError('encoder %r is outside the scoped hot path (SURVEY.md section 2, rows 9-10)' % name)
