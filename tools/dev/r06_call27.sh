#!/bin/bash
# round 6, call 27: the whole-row GroupNorm statistics kernel (128 row ranges, the FPN head's four launches per frame) against the per-group
# kernel (AOT_GN_SPLIT=32): unit tests, goldens, A/B on R50-AOTL and R50-DeAOTL
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
{
timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -k "groupnorm or glue or gn_" 2>&1 | tail -3
timeout 2400 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "(bf16x6 and (c2_r50_aotl_70 or c3b_r50_deaotl_70)) or end_to_end_vs_reference_golden or multi_group" 2>&1 | tail -4
one() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d['config']
print(' value', d['value'], c.get('repeat_fps'), 'single', (c.get('single_stream') or {}).get('fps'), 'jf', {k: v for k, v in (c.get('jf_vs_reference') or {}).items() if k.startswith('pixels')})
PY
}
for m in r50_aotl r50_deaotl; do
B="python bench.py --model $m --steps 60 --warmup 10 --no-other-configs --no-cpu-baseline --no-x6 --no-roofline --no-whole-clip"
for rep in 1 2; do
  echo "== $m whole-row statistics (default), pass $rep"; timeout 900 $B > $O/ab_on.json 2> $O/ab_on.err; one $O/ab_on.json
  echo "== $m AOT_GN_SPLIT=32, pass $rep"; AOT_GN_SPLIT=32 timeout 900 $B > $O/ab_off.json 2> $O/ab_off.err; one $O/ab_off.json
done
done
} > $O/r06_call27.txt 2>&1
cat $O/r06_call27.txt
