#!/bin/bash
# bf16 split-K weight gradients: tests, step time (both precisions), kernel stats of the bf16 step; the 528-launch attention stress test
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_training_backward_gpu.py -x -q -m gpu -k "bf16 or flat_train or train_step_object" 2>&1 | tail -4 > $O/r04_call22_tests.txt
timeout 300 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "reproducible_under_load" 2>&1 | tail -3 >> $O/r04_call22_tests.txt
cat $O/r04_call22_tests.txt
for prec in bf16 f32 bf16; do
  timeout 300 python tools/dev/train_ddp.py --gpus 1 --model r50_deaotl --precision $prec --steps 6 2>/dev/null | tail -n 1 | cut -c1-140
done | tee $O/r04_call22_steps.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof
WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29555 timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o t -- python $R/tools/dev/train_ddp.py --gpus 1 --model r50_deaotl --precision bf16 --steps 3 > $O/r04_train_prof3.json 2> $O/r04_train_prof3.err
python $R/tools/dev/prof_summary.py $(find $O/prof -name "*.db" | head -1) $O/r04_train_step_kernel_stats_bf16_c.txt | head -40 | cut -c1-125
rm -rf $O/prof
