cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "attention or end_to_end_full or free_running_masks or lane_batched or graph_replay" > $O/attntest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/attntest.log
timeout 400 python bench.py --no-cpu-baseline --no-jf --steps 207 --warmup 5 > $O/bench_attn.log 2>&1; tail -1 $O/bench_attn.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['single_stream'], d['roofline'])"
