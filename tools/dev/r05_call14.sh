#!/bin/bash
# round 5, call 14: same box, alternating: 64x64 default = both operands staged (65) / weights direct (66)
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
F="--steps 207 --warmup 10 --no-other-configs --no-cpu-baseline --no-x6 --no-roofline --no-whole-clip --no-jf"
{
for t in 66 65 66 65; do
  echo "== bench, 64x64 default kernel $t"; AOT_X6_DEF64=$t timeout 600 python bench.py $F 2>/dev/null | tail -1 > $O/_b.json
  python - <<'PY'
import json
d = json.loads(open('gpurun_out/_b.json').read()); c = d['config']
print(' value', d['value'], c.get('repeat_fps'), 'single', (c.get('single_stream') or {}).get('fps'), (c.get('single_stream') or {}).get('repeat_fps'))
PY
done
} > $O/r05_call14.txt 2>&1
cat $O/r05_call14.txt
