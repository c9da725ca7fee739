#!/bin/bash
# chunk-wise pack kernels + bf16x6 self-attention: kernel tests, engine goldens, timings, bench lines of the family
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_parity_gpu.py -q --timeout 500 -k "attention_x6 or bf16x6_engine or reproducible_under_load" > gpurun_out/r03t_x6_tests.log 2>&1
tail -3 gpurun_out/r03t_x6_tests.log
timeout 100 python tools/dev/mb_attn_x6.py "" quick 2>&1 | grep -v amdgpu | cut -c1-150
timeout 100 python tools/dev/mb_attn_x6.py "" 2>&1 | grep "pack of\|two lanes"
timeout 100 python tools/dev/mb_gated_x6.py "" quick 2>&1 | grep "pack of"
timeout 300 python bench.py --steps 20 --warmup 5 --mfma bf16x6 --no-cpu-baseline --no-roofline > gpurun_out/r03t_bench20_bf16x6.json 2> gpurun_out/r03t_bench20_bf16x6.err
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r03t_bench20_bf16x6.json').read().strip().splitlines()[-1])
    j=d['config'].get('jf_vs_reference') or {}
    print('bf16x6 bench', d['value'], d['config'].get('repeat_fps'), d['config'].get('single_stream',{}).get('fps'), {k:j.get(k) for k in ('J&F','pixels_differing','pixels_outside_near_ties')})
except Exception as e:
    print('bench failed', e); print(open('gpurun_out/r03t_bench20_bf16x6.err').read()[-1500:])
PY
