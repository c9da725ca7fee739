from .default import preset

ModelConfig = preset('deaots')
