"""Import path of the reference's DeAOT base preset (configs/models/default_deaot.py)."""
from .default import DefaultDeAOTModelConfig as DefaultModelConfig  # noqa: F401
