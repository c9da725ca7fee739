#!/bin/bash
# bias gradient of matmul as a column sum, GEMM entry without the unused operand layout: tests, step time, per-site profile
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_training_backward_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error|assert" | head -20 > $O/r04_call25_tests.txt
cat $O/r04_call25_tests.txt
for prec in bf16 f32; do
  timeout 300 python tools/dev/train_ddp.py --gpus 1 --model r50_deaotl --precision $prec --steps 6 2>/dev/null | tail -n 1 | cut -c1-140
done | tee $O/r04_call25_steps.txt
timeout 400 python tools/dev/train_ddp.py --gpus 1 --model r50_deaotl --precision bf16 --steps 3 --profile $O/r04_train_step_by_site3.txt 2>$O/r04_call25.err | tail -n 1 | cut -c1-140
head -45 $O/r04_train_step_by_site3.txt | cut -c1-150
sed -n '/products on the strided/,$p' $O/r04_train_step_by_site3.txt | head -30
