#!/bin/bash
# bf16x6 attention in the engine: kernel tests, the engine goldens of the family, the kernel timing and the family's bench line
mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_parity_gpu.py -q --timeout 400 -k "attention_x6 or bf16x6_engine or reproducible_under_load" > gpurun_out/r03o_x6_tests.log 2>&1
tail -4 gpurun_out/r03o_x6_tests.log
timeout 200 python tools/dev/mb_attn_x6.py "" > gpurun_out/r03l_mb_attn_x6.txt 2>&1; grep -v amdgpu gpurun_out/r03l_mb_attn_x6.txt | cut -c1-150
timeout 300 python bench.py --steps 20 --warmup 5 --mfma bf16x6 --no-cpu-baseline --no-roofline > gpurun_out/r03o_bench20_bf16x6.json 2> gpurun_out/r03o_bench20_bf16x6.err
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r03o_bench20_bf16x6.json').read().strip().splitlines()[-1])
    print('bf16x6 bench', d['value'], d['dtype'], d['config'].get('repeat_fps'), d['config'].get('single_stream',{}).get('fps'), d['config'].get('jf_vs_reference'))
except Exception as e:
    print('bench failed', e); print(open('gpurun_out/r03o_bench20_bf16x6.err').read()[-1500:])
PY
