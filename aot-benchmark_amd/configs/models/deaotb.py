from .default import preset

ModelConfig = preset('deaotb')
