"""Model registry (the reference's build_vos_model entry point, networks/models/__init__.py:5-11)."""
from networks.models.aot import AOT
from networks.models.deaot import DeAOT

_MODELS = {'aot': AOT, 'deaot': DeAOT}


def build_vos_model(name, cfg, **kwargs):
    try:
        cls = _MODELS[name]
    except KeyError:
        raise NotImplementedError('VOS model %r' % (name,)) from None
    return cls(cfg, encoder=cfg.MODEL_ENCODER, **kwargs)
