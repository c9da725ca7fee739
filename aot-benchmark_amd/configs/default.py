"""Engine config for the inference path.  Mirrors ``DefaultEngineConfig`` of the
reference (configs/default.py:5-107) for the attributes the hot path and its callers
read; unlike the reference it never creates directories (``init_dir`` :109-138 is
training/IO plumbing and out of scope)."""
import importlib


class DefaultEngineConfig():
    def __init__(self, exp_name='default', model='aott'):
        model_cfg = importlib.import_module('configs.models.' + model).ModelConfig()
        self.__dict__.update(model_cfg.__dict__)
        self.EXP_NAME = exp_name + '_' + self.MODEL_NAME
        self.TEST_GPU_ID = 0
        self.TEST_GPU_NUM = 1
        self.TEST_CKPT_PATH = None
        self.TEST_FLIP = False
        self.TEST_MULTISCALE = [1]
        self.TEST_MAX_SHORT_EDGE = None
        self.TEST_MAX_LONG_EDGE = 800 * 1.3
        self.DIST_BACKEND = 'nccl'  # RCCL on ROCm


EngineConfig = DefaultEngineConfig
