"""DeAOTB preset (reference configs/models/deaotb.py)."""
from .default import DefaultDeAOTModelConfig


class ModelConfig(DefaultDeAOTModelConfig):
    def __init__(self):
        super().__init__()
        self.MODEL_NAME = 'DeAOTB'
        self.MODEL_LSTT_NUM = 3
