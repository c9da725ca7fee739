"""AOTS preset (reference configs/models/aots.py)."""
from .default import DefaultModelConfig


class ModelConfig(DefaultModelConfig):
    def __init__(self):
        super().__init__()
        self.MODEL_NAME = 'AOTS'
        self.MODEL_LSTT_NUM = 2
