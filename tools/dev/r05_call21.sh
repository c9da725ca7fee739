#!/bin/bash
# round 5, call 21: every bf16x6 parity cell + the kernel families' tests on the final library, then the driver's command (record)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -k "bf16x6 or c4 or gn_partials or frame_tail or phase_shifted" 2>&1 | tail -6 > $O/r05_final_cells.txt
cat $O/r05_final_cells.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05z2_bench20.json 2> $O/r05z2_bench20.err
tail -1 $O/r05z2_bench20.json | cut -c1-400
grep -i "error\|Traceback" $O/r05z2_bench20.err | head -5
