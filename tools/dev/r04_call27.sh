#!/bin/bash
# The N > 1 control flow of bench.py WITH device work on the one-GPU box: two ranks share GPU 0, collectives over gloo
# (barriers, MAX reductions, the stats gather, rank-0-only legs skipped at N > 1).  The numbers mean nothing; the run must
# finish, print ONE line from rank 0 with n_gpus 2, and both ranks must report the same collective sequence.
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 600 python bench.py --gpus 2 --share-gpu --steps 20 --warmup 5 --repeats 2 > $O/r04_share_gpu_world2.json 2> $O/r04_share_gpu_world2.err
echo rc $?
wc -l $O/r04_share_gpu_world2.json
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r04_share_gpu_world2.json').read().strip().splitlines()[-1])
c = d['config']
print('n_gpus', d['n_gpus'], 'value', d['value'], 'data', d['data'][:60])
print('whole_clip', c.get('whole_clip', {}).get('fps'), 'single', c.get('single_stream', {}).get('fps'), 'host', c.get('host'))
print('other_configs' in c, d.get('cpu_baseline'))
PY
tail -5 $O/r04_share_gpu_world2.err
