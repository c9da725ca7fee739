#!/bin/bash
# round 4, call 3: ISA-level bisect of the two failing instruction orders; TrainStep over RCCL (world 1), train_ddp.py; config 3 free-running
# (one cell); the driver's bench line with the other_configs legs
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 200 tools/dev/x6_hazard_mod tools/dev/bisect/KERNEL tools/dev/bisect/e*.co > $O/r04_isa_bisect_pad2.txt 2>&1; echo "bisect rc $?"
timeout 200 tools/dev/x6_hazard_mod tools/dev/bisect8/KERNEL tools/dev/bisect8/e*.co > $O/r04_isa_bisect_pad8.txt 2>&1; echo "bisect8 rc $?"
cat $O/r04_isa_bisect_pad2.txt | cut -c1-220
timeout 600 python -m pytest tests/test_training_backward_gpu.py -x -q -m gpu -k "nccl_world1" -s > $O/r04_train_tests3.txt 2>&1; echo "train tests rc $?"; tail -n 6 $O/r04_train_tests3.txt
for prec in f32 bf16; do
  timeout 300 python tools/dev/train_ddp.py --gpus 1 --model r50_deaotl --precision $prec --steps 4 > $O/r04_train_ddp_$prec.json 2> $O/r04_train_ddp_$prec.err; echo "train_ddp $prec rc $?"; tail -n 2 $O/r04_train_ddp_$prec.json; grep -i "error" $O/r04_train_ddp_$prec.err | tail -n 3
done
timeout 300 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "free_running and c3_swinb_deaotl_480_70 and throughput" > $O/r04_swin_free.txt 2>&1; echo "swin free-running rc $?"; tail -n 3 $O/r04_swin_free.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04_bench20.json 2> $O/r04_bench20.err; echo "bench rc $?"; tail -n 14 $O/r04_bench20.err
