"""DeAOTL preset (reference configs/models/deaotl.py)."""
from .default import DefaultDeAOTModelConfig


class ModelConfig(DefaultDeAOTModelConfig):
    def __init__(self):
        super().__init__()
        self.MODEL_NAME = 'DeAOTL'
        self.MODEL_LSTT_NUM = 3
        self.TRAIN_LONG_TERM_MEM_GAP = 2
        self.TEST_LONG_TERM_MEM_GAP = 5
