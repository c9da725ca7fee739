#!/bin/bash
# round 6, call 11: the launch-floor fusions (LayerNorm prologue, GN statistics out of the split-K reduce, GN-apply inside the upsampling and
# inside conv_out, per-wave partial reduction of the fused GN + GELU + dw5x5): unit tests, the engine goldens in the bench's
# configurations, then alternating A/B bench runs (fusions on / off through the AOT_NO_* switches)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
{
timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "layernorm_linear or gn_statistics_from_splitk or gn_bilinear or gn_conv1x1 or gn_partials or merged_qkv or layernorm_groupnorm or bilinear" 2>&1 | tail -8
timeout 2400 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "(bf16x6 and c2_r50_aotl_70) or end_to_end_vs_reference_golden or multi_group or graph_replay or demo_real or sequence_evaluator" 2>&1 | tail -8
B="python bench.py --steps 60 --warmup 10 --no-other-configs --no-cpu-baseline --no-x6 --no-roofline --no-jf --no-whole-clip"
one() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d['config']
print(' value', d['value'], c.get('repeat_fps'), 'single', (c.get('single_stream') or {}).get('fps'), (c.get('single_stream') or {}).get('repeat_fps'))
PY
}
for rep in 1 2; do
  echo "== fused (default), pass $rep"; timeout 600 $B > $O/ab_on.json 2> $O/ab_on.err; one $O/ab_on.json
  echo "== unfused (AOT_NO_LN_FUSE AOT_NO_GNR_FUSE AOT_NO_GN_UP), pass $rep"; AOT_NO_LN_FUSE=1 AOT_NO_GNR_FUSE=1 AOT_NO_GN_UP=1 timeout 600 $B > $O/ab_off.json 2> $O/ab_off.err; one $O/ab_off.json
done
echo "== only LN off"; AOT_NO_LN_FUSE=1 timeout 600 $B > $O/ab_x.json 2> $O/ab_x.err; one $O/ab_x.json
echo "== only GNR off"; AOT_NO_GNR_FUSE=1 timeout 600 $B > $O/ab_x.json 2> $O/ab_x.err; one $O/ab_x.json
echo "== only GN_UP off"; AOT_NO_GN_UP=1 timeout 600 $B > $O/ab_x.json 2> $O/ab_x.err; one $O/ab_x.json
} > $O/r06_call11.txt 2>&1
cat $O/r06_call11.txt
