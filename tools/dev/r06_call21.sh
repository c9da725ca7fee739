#!/bin/bash
# round 6, call 21: lgp_aggregate_kernel with 16-channel chunks (three workgroups per CU) against 32 (one)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
L=$R/aot-benchmark_amd/csrc/libaot_hip_cb16.so
{
for i in 1 2; do
echo "== CB 32 (product)"; python tools/dev/mb_local_gated.py "" 2>&1 | grep -v amdgpu.ids; python tools/dev/mb_local_gated.py "" 30 53 2>&1 | grep -v amdgpu.ids
echo "== CB 16"; python tools/dev/mb_local_gated.py $L 2>&1 | grep -v amdgpu.ids; python tools/dev/mb_local_gated.py $L 30 53 2>&1 | grep -v amdgpu.ids
done
AOT_HIP_LIB=$L timeout 600 python -m pytest tests/test_parity_gpu.py -q -m gpu -k "local_gated" 2>&1 | tail -3
one() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d['config']
print(' value', d['value'], c.get('repeat_fps'), 'single', (c.get('single_stream') or {}).get('fps'), 'jf', {k: v for k, v in (c.get('jf_vs_reference') or {}).items() if k.startswith('pixels')})
PY
}
B="python bench.py --model r50_deaotl --steps 60 --warmup 10 --no-other-configs --no-cpu-baseline --no-x6 --no-roofline --no-whole-clip"
for rep in 1 2; do
  echo "== r50_deaotl CB 32, pass $rep"; timeout 900 $B > $O/ab_on.json 2> $O/ab_on.err; one $O/ab_on.json
  echo "== r50_deaotl CB 16, pass $rep"; AOT_HIP_LIB=$L timeout 900 $B > $O/ab_off.json 2> $O/ab_off.err; one $O/ab_off.json
done
} > $O/r06_call21.txt 2>&1
cat $O/r06_call21.txt
