"""Model factory (reference networks/models/__init__.py:5-11)."""
from networks.models.aot import AOT


def build_vos_model(name, cfg, **kwargs):
    if name == 'aot':
        return AOT(cfg, encoder=cfg.MODEL_ENCODER, **kwargs)
    if name == 'deaot':
        raise NotImplementedError('DeAOT (gated propagation) is the next row of the scope table; '
                                  'its CPU oracle exists in oracle/aot_oracle.py but the HIP kernels are not built yet')
    raise NotImplementedError
