#!/bin/bash
# round 4, call 2: trans-op hazard probe + alpha-padding variants of the reproducer; the new training pieces (bf16 products, flat
# training state, TrainStep over a one-rank RCCL group, train_ddp.py); bench line with phase timings
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 120 tools/dev/trans_hazard_probe > $O/r04_trans_hazard_probe.txt 2>&1; echo "trans probe rc $?"
timeout 200 tools/dev/x6_hazard 30 > $O/r04_x6_hazard_b.txt 2>&1; echo "x6_hazard rc $?"; grep -c differing $O/r04_x6_hazard_b.txt
timeout 600 python -m pytest tests/test_training_backward_gpu.py -x -q -m gpu -k "gemm_bf16 or linear_bf16 or flat_train_state or bf16_close or nccl_world1" -s > $O/r04_train_tests2.txt 2>&1; echo "train tests rc $?"; tail -n 12 $O/r04_train_tests2.txt
for prec in f32 bf16; do
  timeout 300 python tools/dev/train_ddp.py --gpus 1 --model r50_deaotl --precision $prec --steps 4 > $O/r04_train_ddp_$prec.json 2> $O/r04_train_ddp_$prec.err; echo "train_ddp $prec rc $?"; tail -n 2 $O/r04_train_ddp_$prec.json; tail -n 3 $O/r04_train_ddp_$prec.err
done
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs > $O/r04_bench20.json 2> $O/r04_bench20.err; echo "bench rc $?"; tail -n 12 $O/r04_bench20.err
