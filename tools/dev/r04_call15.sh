#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_training_backward_gpu.py tests/test_training_gpu.py -x -q -m gpu > $O/r04_train_tests4.txt 2>&1; echo "train tests rc $?"; tail -n 5 $O/r04_train_tests4.txt
for prec in f32 bf16; do
  timeout 300 python tools/dev/train_ddp.py --gpus 1 --model r50_deaotl --precision $prec --steps 4 > $O/r04_train_ddp2_$prec.json 2> $O/r04_train_ddp2_$prec.err; echo "train_ddp $prec rc $?"; tail -n 1 $O/r04_train_ddp2_$prec.json | cut -c1-330; grep -i "error" $O/r04_train_ddp2_$prec.err | tail -n 3
done
