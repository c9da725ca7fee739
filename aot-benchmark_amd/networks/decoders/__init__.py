"""Decoder registry (the reference's build_decoder entry point, networks/decoders/__init__.py)."""
from networks.decoders.fpn import FPNSegmentationHead

_DECODERS = {'fpn': FPNSegmentationHead}


def build_decoder(name, **kwargs):
    try:
        cls = _DECODERS[name]
    except KeyError:
        raise NotImplementedError('decoder %r' % (name,)) from None
    return cls(**kwargs)
