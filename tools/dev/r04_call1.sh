#!/bin/bash
# round 4, call 1: MFMA hazard probes (stand-alone binaries), Swin goldens after the trunk calibration (free-running without tie-sync),
# the new training test, bench line with the whole_clip / other_configs legs
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 120 tools/dev/mfma_hazard_probe > $O/r04_mfma_hazard_probe.txt 2>&1; echo "probe rc $?"
timeout 300 tools/dev/x6_hazard 40 > $O/r04_x6_hazard.txt 2>&1; echo "x6_hazard rc $?"
tail -n 60 $O/r04_x6_hazard.txt
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "swin" > $O/r04_swin_tests.txt 2>&1; echo "swin tests rc $?"; tail -n 5 $O/r04_swin_tests.txt
timeout 600 python -m pytest tests/test_training_backward_gpu.py -x -q -m gpu -k "freezes or (matches_reference and swinb)" > $O/r04_train_tests.txt 2>&1; echo "train tests rc $?"; tail -n 5 $O/r04_train_tests.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04_bench20.json 2> $O/r04_bench20.err; echo "bench rc $?"; tail -n 8 $O/r04_bench20.err

