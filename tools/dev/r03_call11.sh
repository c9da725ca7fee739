#!/bin/bash
# encoder look-ahead depth sweep (frames of a clip encoded as one batch): 3 (default) / 5 / 8, driver form
mkdir -p gpurun_out
for k in 3 5 8; do
  timeout 200 python bench.py --steps 20 --warmup 5 --encode-ahead $k --no-cpu-baseline --no-jf --no-x6 --no-roofline > gpurun_out/r03k_ahead_$k.json 2> gpurun_out/r03k_ahead_$k.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r03k_ahead_$k.json').read().strip().splitlines()[-1])
    print('ahead', $k, d['value'], d['config'].get('repeat_fps'), d['config'].get('single_stream'))
except Exception as e:
    print('ahead', $k, 'failed', e); print(open('gpurun_out/r03k_ahead_$k.err').read()[-800:])
PY
done
