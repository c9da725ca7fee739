#!/bin/bash
# round 6, call 5: where do the engine's free-running tie flips sit on the fp64 reference?  (three 70-frame clips, bf16x6, timed cell)
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 2400 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "free_running and bf16x6 and throughput and tail" 2>&1 | tail -6
python - <<'PY'
import json
d = json.load(open('gpurun_out/parity_r06.json'))
for case, modes in d.items():
    for mode, rec in modes.items():
        print(case, mode, rec.get('pixels_differing'), rec.get('flips_on_the_fp64_reference'), rec.get('reference_fp32_vs_fp64'))
PY
