#!/bin/bash
# bf16x6 gated attention in the DeAOT engines: kernel tests, the engine goldens of the family, bench lines of the DeAOT models
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_parity_gpu.py -q --timeout 500 -k "attention_x6 or bf16x6_engine or reproducible_under_load" > gpurun_out/r03s_x6_tests.log 2>&1
tail -4 gpurun_out/r03s_x6_tests.log
for m in r50_deaotl swinb_deaotl; do
  timeout 300 python bench.py --model $m --steps 20 --warmup 5 --mfma bf16x6 --no-cpu-baseline --no-roofline > gpurun_out/r03s_bench20_${m}_bf16x6.json 2> gpurun_out/r03s_bench20_${m}_bf16x6.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r03s_bench20_${m}_bf16x6.json').read().strip().splitlines()[-1])
    j=d['config'].get('jf_vs_reference') or {}
    print('$m bf16x6 bench', d['value'], d['dtype'], d['config'].get('repeat_fps'), d['config'].get('single_stream',{}).get('fps'), {k:j.get(k) for k in ('J&F','pixels_differing','pixels_outside_near_ties')})
except Exception as e:
    print('bench failed', e); print(open('gpurun_out/r03s_bench20_${m}_bf16x6.err').read()[-1500:])
PY
done
