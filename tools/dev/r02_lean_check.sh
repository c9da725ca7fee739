cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "conv2d or linear or end_to_end or free_running_masks or multi_group or graph_replay or swin_encoder" > $O/leantest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/leantest.log
timeout 300 python bench.py --no-cpu-baseline --no-jf --steps 207 --warmup 5 > $O/bench_lean.log 2>&1; tail -1 $O/bench_lean.log | cut -c1-1500
