from .default import preset

ModelConfig = preset('aott')
