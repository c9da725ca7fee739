#!/bin/bash
# round 5, call 13: bench with the direct-weight kernel as the family's default; clip streams per GPU 2 / 3 / 4
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
F="--steps 207 --warmup 10 --no-other-configs --no-cpu-baseline --no-x6 --no-roofline --no-whole-clip"
{
for s in 3 2 4; do
  echo "== bench, $s clip streams"; timeout 600 python bench.py $F --streams $s $( [ $s != 3 ] && echo --no-jf ) 2>/dev/null | tail -1 > $O/_b.json
  python - <<'PY'
import json
d = json.loads(open('gpurun_out/_b.json').read()); c = d['config']
print(' value', d['value'], c.get('repeat_fps'), 'single', (c.get('single_stream') or {}).get('fps'), 'mem', c.get('peak_mem_gib'),
      'jf', {k: (c.get('jf_vs_reference') or {}).get(k) for k in ('pixels_differing', 'pixels_outside_near_ties')})
PY
done
for m in r50_deaotl swinb_deaotl; do
  echo "== bench, $m"; timeout 600 python bench.py $F --model $m 2>/dev/null | tail -1 > $O/_b.json
  python - <<'PY'
import json
d = json.loads(open('gpurun_out/_b.json').read()); c = d['config']
print(' value', d['value'], c.get('repeat_fps'), 'single', (c.get('single_stream') or {}).get('fps'),
      'jf', {k: (c.get('jf_vs_reference') or {}).get(k) for k in ('pixels_differing', 'pixels_outside_near_ties')})
PY
done
} > $O/r05_call13.txt 2>&1
cat $O/r05_call13.txt
