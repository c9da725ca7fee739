"""Import path of the reference's DeAOT engines (networks/engines/deaot_engine.py:9-98).  The classes live next to the AOT
engines: the state machine is the same, the DeAOT memory layout is handled by the model (see aot_engine.DeAOTEngine)."""
from networks.engines.aot_engine import DeAOTEngine, DeAOTInferEngine  # noqa: F401
