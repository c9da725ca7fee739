#!/bin/bash
# round 6, call 29: the bench's J&F leg with the boundary measure restated after the DAVIS toolkit (seg2bmap + disk dilation)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python bench.py --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline > $O/r06_jf_check.json 2> $O/r06_jf_check.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_jf_check.json').read().strip().splitlines()[-1]); c=d['config']
print('value', d['value'], 'single', c['single_stream']['fps'])
print('jf', {k: v for k, v in c['jf_vs_reference'].items() if k in ('J','F','J&F','frames','pixels_differing','pixels_outside_near_ties')})
print('fp32 jf', {k: v for k, v in (c['fp32_exact'].get('jf_vs_reference') or {}).items() if k in ('J','F','J&F','pixels_differing','pixels_outside_near_ties')})
PY
grep "J&F pass" $O/r06_jf_check.err
