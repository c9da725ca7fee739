#!/bin/bash
# glue of the training graph: Bernoulli masks in two launches, split nodes instead of slice pairs -- tests, step time; bench sanity
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_training_backward_gpu.py tests/test_training_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error|assert" | head -20 > $O/r04_call28_tests.txt
cat $O/r04_call28_tests.txt
for prec in bf16 f32 bf16; do
  timeout 300 python tools/dev/train_ddp.py --gpus 1 --model r50_deaotl --precision $prec --steps 6 2>/dev/null | tail -n 1 | cut -c1-140
done | tee $O/r04_call28_steps.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench value', d['value'], d['config']['repeat_fps'], 'whole', d['config']['whole_clip']['fps'], 'jf outside', d['config']['jf_vs_reference']['pixels_outside_near_ties'])"
