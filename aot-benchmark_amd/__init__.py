"""MI355X-native AOT/DeAOT inference engine (drop-in for the per-frame path of
yoxu515/aot-benchmark).  Put this directory on ``sys.path``; the importable
packages inside mirror the reference's own layout so caller code keeps working:

    from networks.models import build_vos_model      # reference networks/models/__init__.py:5-11
    from networks.engines import build_engine        # reference networks/engines/__init__.py:5-21

Compute runs in hand-written gfx950 HIP kernels (``csrc/``) behind the C ABI
declared in ``include/aot_hip.h``; there is no CPU fallback.
"""
