#!/bin/bash
# round 5, call 17: is the three-stream figure host-bound?  Two processes x 2 streams and two processes x 3 streams on ONE GPU
# (--share-gpu, gloo) against one process x 3 streams
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
F="--steps 207 --warmup 10 --no-other-configs --no-cpu-baseline --no-x6 --no-roofline --no-whole-clip --no-jf"
P() { python - <<'PY'
import json
d = json.loads(open('gpurun_out/_b.json').read().strip().splitlines()[-1]); c = d['config']
print(' value', d['value'], c.get('repeat_fps'), 'n_gpus', d['n_gpus'], 'streams', c.get('streams_per_gpu'))
PY
}
{
echo "== 1 process x 3 streams"; timeout 600 python bench.py $F 2>/dev/null | tail -1 > $O/_b.json; P
for s in 2 3; do
  echo "== 2 processes x $s streams on one GPU"
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --share-gpu $F --streams $s 2>/dev/null | tail -1 > $O/_b.json; P
done
} > $O/r05_hostbound.txt 2>&1
cat $O/r05_hostbound.txt
