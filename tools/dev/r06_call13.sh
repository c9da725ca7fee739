#!/bin/bash
# round 6, call 13: grouped linear launches, overlapped look-ahead encoding, the tree after the removal of the GN-statistics-in-reduce
# path: unit tests, goldens, A/B of AOT_NO_GROUP and of --overlap-encode on R50-AOTL and SwinB-DeAOTL
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
{
timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "linear_group or layernorm_linear or gn_bilinear or gn_conv1x1 or linear_with_layernorm or gn_partials or encode_ahead" 2>&1 | tail -6
timeout 2400 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "(bf16x6 and (c2_r50_aotl_70 or c3b_r50_deaotl_70)) or end_to_end_vs_reference_golden or multi_group or graph_replay" 2>&1 | tail -6
one() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d['config']
print(' value', d['value'], c.get('repeat_fps'), 'single', (c.get('single_stream') or {}).get('fps'), (c.get('single_stream') or {}).get('repeat_fps'))
PY
}
for m in r50_aotl swinb_deaotl; do
  B="python bench.py --model $m --steps 60 --warmup 10 --no-other-configs --no-cpu-baseline --no-x6 --no-roofline --no-jf --no-whole-clip"
  for rep in 1 2; do
    echo "== $m default, pass $rep"; timeout 600 $B > $O/ab_on.json 2> $O/ab_on.err; one $O/ab_on.json
    echo "== $m AOT_NO_GROUP, pass $rep"; AOT_NO_GROUP=1 timeout 600 $B > $O/ab_off.json 2> $O/ab_off.err; one $O/ab_off.json
    echo "== $m --overlap-encode 1, pass $rep"; timeout 600 $B --overlap-encode 1 > $O/ab_ov.json 2> $O/ab_ov.err; one $O/ab_ov.json
  done
done
echo "== r50_aotl --overlap-encode 1 --encode-ahead 6"; timeout 600 python bench.py --steps 60 --warmup 10 --no-other-configs --no-cpu-baseline --no-x6 --no-roofline --no-jf --no-whole-clip --overlap-encode 1 --encode-ahead 6 > $O/ab_ov.json 2> $O/ab_ov.err; one $O/ab_ov.json
tail -5 $O/ab_ov.err
} > $O/r06_call13.txt 2>&1
cat $O/r06_call13.txt
