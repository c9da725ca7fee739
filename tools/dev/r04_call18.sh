#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
for s in 3 4 5 6; do
  timeout 300 python bench.py --gpus 1 --steps 207 --warmup 5 --streams $s --no-other-configs --no-cpu-baseline --no-x6 --no-jf --no-roofline --repeats 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams', d['config']['streams_per_gpu'], 'fps', d['value'], d['config']['repeat_fps'], 'single', d['config']['single_stream']['fps'], 'mem', d['config']['peak_mem_gib'])"
done
