#!/bin/bash
# round 6, call 23: LDS-tiled 5x5 depthwise convolution for the GPM blocks' tails against the per-tap kernel (AOT_NO_DW_TILED)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
{
for i in 1 2; do
echo "== lgp CB 8 (product)"; python tools/dev/mb_local_gated.py "" 2>&1 | grep -v amdgpu.ids
echo "== lgp CB 4"; python tools/dev/mb_local_gated.py $R/aot-benchmark_amd/csrc/libaot_hip_cb4.so 2>&1 | grep -v amdgpu.ids
done
timeout 600 python -m pytest tests/test_parity_gpu.py -q -m gpu -k "dwconv or local_gated" 2>&1 | tail -3
timeout 2400 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "(bf16x6 and c3b_r50_deaotl_70) or deaot" 2>&1 | tail -4
one() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d['config']
print(' value', d['value'], c.get('repeat_fps'), 'single', (c.get('single_stream') or {}).get('fps'), 'jf', {k: v for k, v in (c.get('jf_vs_reference') or {}).items() if k.startswith('pixels')})
PY
}
for m in r50_deaotl swinb_deaotl; do
B="python bench.py --model $m --steps 60 --warmup 10 --no-other-configs --no-cpu-baseline --no-x6 --no-roofline --no-whole-clip"
for rep in 1 2; do
  echo "== $m tiled (default), pass $rep"; timeout 900 $B > $O/ab_on.json 2> $O/ab_on.err; one $O/ab_on.json
  echo "== $m AOT_NO_DW_TILED, pass $rep"; AOT_NO_DW_TILED=1 timeout 900 $B > $O/ab_off.json 2> $O/ab_off.err; one $O/ab_off.json
done
done
} > $O/r06_call23.txt 2>&1
cat $O/r06_call23.txt
