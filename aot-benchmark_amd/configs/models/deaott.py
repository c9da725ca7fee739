"""DeAOTT preset (reference configs/models/deaott.py)."""
from .default import DefaultDeAOTModelConfig


class ModelConfig(DefaultDeAOTModelConfig):
    def __init__(self):
        super().__init__()
        self.MODEL_NAME = 'DeAOTT'
