#!/bin/bash
# round 6, call 24: the GPM blocks' SELF gated propagation on the bf16x6 64-query kernel (frame's own K / V packed into a one-frame bank)
# against the fp32 kernel (AOT_NO_SELF_X6): DeAOT goldens incl. the 70-frame Swin-B clip, then A/B on both DeAOT configurations
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
{
timeout 3000 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "(bf16x6 and (c3b_r50_deaotl_70 or c3_swinb_deaotl_480)) or deaot" 2>&1 | tail -5
one() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d['config']
print(' value', d['value'], c.get('repeat_fps'), 'single', (c.get('single_stream') or {}).get('fps'), 'jf', {k: v for k, v in (c.get('jf_vs_reference') or {}).items() if k.startswith('pixels')})
PY
}
for m in r50_deaotl swinb_deaotl; do
B="python bench.py --model $m --steps 60 --warmup 10 --no-other-configs --no-cpu-baseline --no-x6 --no-roofline --no-whole-clip"
for rep in 1 2; do
  echo "== $m self propagation on x6 (default), pass $rep"; timeout 900 $B > $O/ab_on.json 2> $O/ab_on.err; one $O/ab_on.json
  echo "== $m AOT_NO_SELF_X6, pass $rep"; AOT_NO_SELF_X6=1 timeout 900 $B > $O/ab_off.json 2> $O/ab_off.err; one $O/ab_off.json
done
done
} > $O/r06_call24.txt 2>&1
cat $O/r06_call24.txt
