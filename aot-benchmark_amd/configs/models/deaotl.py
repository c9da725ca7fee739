from .default import preset

ModelConfig = preset('deaotl')
