from .default import preset

ModelConfig = preset('swinb_aotl')
