from .default import preset

ModelConfig = preset('r101_aotl')
