#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
rm -rf $O/prof
WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29555 timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o t -- python $R/tools/dev/train_ddp.py --gpus 1 --model r50_deaotl --precision bf16 --steps 3 > $O/r04_train_prof2.json 2> $O/r04_train_prof2.err
python $R/tools/dev/prof_summary.py $(find $O/prof -name "*.db" | head -1) $O/r04_train_step_kernel_stats_bf16_b.txt | head -45 | cut -c1-125
rm -rf $O/prof
