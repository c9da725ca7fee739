from .default import preset

ModelConfig = preset('aotb')
