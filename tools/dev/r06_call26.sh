#!/bin/bash
# round 6, call 26: register caps for two vector kernels -- swin_window_attn_kernel at three waves per SIMD (168 registers instead of 254),
# gn_act_dwconv5_kernel at four (122 instead of 154) -- against the library before (libaot_hip_old.so); bit-identical by construction
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
L=$R/aot-benchmark_amd/csrc/libaot_hip_old.so
{
timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -k "swin or dwconv or gn_partials or layernorm_groupnorm" 2>&1 | tail -3
one() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d['config']
print(' value', d['value'], c.get('repeat_fps'), 'single', (c.get('single_stream') or {}).get('fps'), 'jf', {k: v for k, v in (c.get('jf_vs_reference') or {}).items() if k.startswith('pixels')})
PY
}
for m in swinb_deaotl r50_aotl; do
B="python bench.py --model $m --steps 60 --warmup 10 --no-other-configs --no-cpu-baseline --no-x6 --no-roofline --no-whole-clip"
for rep in 1 2; do
  echo "== $m register caps (product), pass $rep"; timeout 900 $B > $O/ab_on.json 2> $O/ab_on.err; one $O/ab_on.json
  echo "== $m library before, pass $rep"; AOT_HIP_LIB=$L timeout 900 $B > $O/ab_off.json 2> $O/ab_off.err; one $O/ab_off.json
done
done
} > $O/r06_call26.txt 2>&1
cat $O/r06_call26.txt
