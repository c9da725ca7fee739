#!/bin/bash
# the bf16 tests again (summary line kept this time) + the device time of one training step by launching site
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_training_backward_gpu.py -x -q -m gpu -k "bf16 or flat_train or train_step_object" 2>&1 | grep -E "passed|failed|error|Error|assert" | head -20 > $O/r04_call23_tests.txt
cat $O/r04_call23_tests.txt
timeout 400 python tools/dev/train_ddp.py --gpus 1 --model r50_deaotl --precision bf16 --steps 3 --profile $O/r04_train_step_by_site.txt 2>$O/r04_call23.err | tail -n 1 | cut -c1-140
head -70 $O/r04_train_step_by_site.txt | cut -c1-170
