from .default import preset

ModelConfig = preset('aots')
