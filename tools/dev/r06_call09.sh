#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "demo_real_images" 2>&1 | tail -15
timeout 3000 python -m pytest tests/test_parity_gpu.py -q -m gpu -k "free_running_masks_equal_reference and (c3b or swinb)" 2>&1 | tail -8
python - <<'PY'
import json
d = json.load(open('gpurun_out/parity_r06.json'))
for k, v in d.items():
    if 'demo' in k: print(k, json.dumps(v)[:1500])
PY
