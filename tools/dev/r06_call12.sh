#!/bin/bash
# round 6, call 12: second form of the fusions (LayerNorm statistics along the k-loop, GN last-arriver over all groups at once, LayerNorm
# output from linear2's reduce): unit tests, engine goldens, alternating A/B bench runs; split-K sweep of the LSTT linears
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
{
timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "layernorm_linear or gn_statistics_from_splitk or gn_bilinear or gn_conv1x1 or gn_partials or merged_qkv or linear_with_layernorm" 2>&1 | tail -8
timeout 2400 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "(bf16x6 and c2_r50_aotl_70) or end_to_end_vs_reference_golden or multi_group or graph_replay or demo_real" 2>&1 | tail -8
B="python bench.py --steps 60 --warmup 10 --no-other-configs --no-cpu-baseline --no-x6 --no-roofline --no-jf --no-whole-clip"
one() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d['config']
print(' value', d['value'], c.get('repeat_fps'), 'single', (c.get('single_stream') or {}).get('fps'), (c.get('single_stream') or {}).get('repeat_fps'))
PY
}
for rep in 1 2; do
  echo "== fused (default), pass $rep"; timeout 600 $B > $O/ab_on.json 2> $O/ab_on.err; one $O/ab_on.json
  echo "== unfused (all AOT_NO_* switches), pass $rep"; AOT_NO_LN_FUSE=1 AOT_NO_GNR_FUSE=1 AOT_NO_GN_UP=1 AOT_NO_LNO_FUSE=1 timeout 600 $B > $O/ab_off.json 2> $O/ab_off.err; one $O/ab_off.json
done
echo "== only LN off"; AOT_NO_LN_FUSE=1 timeout 600 $B > $O/ab_x.json 2> $O/ab_x.err; one $O/ab_x.json
echo "== only GNR off"; AOT_NO_GNR_FUSE=1 timeout 600 $B > $O/ab_x.json 2> $O/ab_x.err; one $O/ab_x.json
echo "== only LNO off"; AOT_NO_LNO_FUSE=1 timeout 600 $B > $O/ab_x.json 2> $O/ab_x.err; one $O/ab_x.json
echo "== split-K sweep of the LSTT linears, one lane (x6 = the dispatch; x6kN = N slices on the 64x64 direct-weight kernel)"
timeout 300 python tools/dev/mb_gemm.py x6,x6d,x6k2,x6k4,x6k8 "" lstt 1
} > $O/r06_call12.txt 2>&1
cat $O/r06_call12.txt
