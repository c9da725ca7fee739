#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
for prec in f32 bf16 f32; do
  timeout 300 python tools/dev/train_ddp.py --gpus 1 --model r50_deaotl --precision $prec --steps 6 2>/dev/null | tail -n 1 | cut -c1-120
done
