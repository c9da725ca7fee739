from .default import preset

ModelConfig = preset('deaott')
