#!/bin/bash
# round 6, call 8: the real-image end-to-end test, then the whole free-running matrix with the fp64-reference assertions
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "demo_real_images" 2>&1 | tail -15
timeout 3000 python -m pytest tests/test_parity_gpu.py -q -m gpu -k "free_running_masks_equal_reference" 2>&1 | tail -15
