from .default import preset

ModelConfig = preset('aotl')
