"""Model factory (reference networks/models/__init__.py:5-11)."""
from networks.models.aot import AOT
from networks.models.deaot import DeAOT


def build_vos_model(name, cfg, **kwargs):
    if name == 'aot':
        return AOT(cfg, encoder=cfg.MODEL_ENCODER, **kwargs)
    if name == 'deaot':
        return DeAOT(cfg, encoder=cfg.MODEL_ENCODER, **kwargs)
    raise NotImplementedError
