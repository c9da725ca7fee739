"""DeAOTS preset (reference configs/models/deaots.py)."""
from .default import DefaultDeAOTModelConfig


class ModelConfig(DefaultDeAOTModelConfig):
    def __init__(self):
        super().__init__()
        self.MODEL_NAME = 'DeAOTS'
        self.MODEL_LSTT_NUM = 2
