#!/bin/bash
# round 6, call 20: the driver's command after the single-stream leg moved to whole clips (two runs)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
for i in 1 2; do
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06y${i}_bench20.json 2> $O/r06y${i}_bench20.err
python - $O/r06y${i}_bench20.json <<'PY'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d['config']
print('value',d['value'],c['repeat_fps'],'whole',c['whole_clip']['fps'],'single',{k:c['single_stream'].get(k) for k in ('fps','repeat_fps','frames','windows_fps','encode_overlapped')},'online',c['single_stream'].get('online',{}).get('fps'))
for k,v in c['other_configs'].items(): print('  ',k, v.get('fps'), v.get('whole_clip_fps'), v.get('single_stream_fps'))
PY
done
