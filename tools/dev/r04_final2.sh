#!/bin/bash
# second closing pass (the library gained the experimental pre-split GEMM member after the first): the whole GPU suite, a short bench
# line, the training step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -6 > $O/r04z2_gpu_suite.txt
cat $O/r04z2_gpu_suite.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline 2>/dev/null | tail -n 1 > $O/r04z2_bench20_short.json
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r04z2_bench20_short.json').read().strip().splitlines()[-1])
c = d['config']
print('value', d['value'], c['repeat_fps'], 'whole', c['whole_clip']['fps'], 'single', c['single_stream']['fps'], 'fp32', c['fp32_exact']['value'], 'jf outside', c['jf_vs_reference']['pixels_outside_near_ties'], 'roofline', d['roofline']['frac'])
PY
timeout 300 python tools/dev/train_ddp.py --gpus 1 --model r50_deaotl --precision bf16 --steps 6 2>/dev/null | tail -n 1 | cut -c1-150
