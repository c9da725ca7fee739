"""Engine factory (reference networks/engines/__init__.py:5-21)."""
from networks.engines.aot_engine import AOTEngine, AOTInferEngine, DeAOTEngine, DeAOTInferEngine


def build_engine(name, phase='train', **kwargs):
    # (kwargs may also carry long_term_mem_max, this repo's bounded-bank extension; default None = reference behaviour)
    if name == 'aotengine':
        if phase == 'eval':
            return AOTInferEngine(**kwargs)
        raise NotImplementedError("phase %r: only the inference engine ('eval') is on the scoped hot path" % phase)
    if name == 'deaotengine':
        if phase == 'eval':
            return DeAOTInferEngine(**kwargs)
        raise NotImplementedError("phase %r: only the inference engine ('eval') is on the scoped hot path" % phase)
    raise NotImplementedError
