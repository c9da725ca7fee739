from .default import preset

ModelConfig = preset('r50_aotl')
