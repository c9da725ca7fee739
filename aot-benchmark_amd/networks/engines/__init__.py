"""Engine factory (reference networks/engines/__init__.py:5-21)."""
from networks.engines.aot_engine import AOTEngine, AOTInferEngine, DeAOTEngine, DeAOTInferEngine

_ENGINES = {'aotengine': {'train': AOTEngine, 'eval': AOTInferEngine},
            'deaotengine': {'train': DeAOTEngine, 'eval': DeAOTInferEngine}}


def build_engine(name, phase='train', **kwargs):
    """'eval': the caller-facing inference engine; 'train': the single-group engine whose forward() is the training step's
    forward (kwargs may also carry long_term_mem_max, this repo's bounded-bank extension; default None = reference
    behaviour)."""
    try:
        return _ENGINES[name][phase](**kwargs)
    except KeyError:
        raise NotImplementedError('engine %r, phase %r' % (name, phase)) from None
