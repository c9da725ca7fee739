#!/bin/bash
# round 6, call 15: attn_x6_d32_kernel with a key tile as two phases (vector phase, then 24 MFMAs alternating between the two accumulators)
# against the round-5 step (build switch AOT_X6_OLDSTEP): launch times, unit tests, bench A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
L=$R/aot-benchmark_amd/csrc/libaot_hip_oldstep.so
{
echo "== two phases (product)"; timeout 300 python tools/dev/mb_attn_x6.py "" 2>&1 | grep -v amdgpu.ids
echo "== round-5 step"; timeout 300 python tools/dev/mb_attn_x6.py $L 2>&1 | grep -v amdgpu.ids
echo "== unit tests"
timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -k "attention_x6 or attention_kernels_reproducible" 2>&1 | tail -5
one() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d['config']
print(' value', d['value'], c.get('repeat_fps'), 'single', (c.get('single_stream') or {}).get('fps'), 'jf', {k: v for k, v in (c.get('jf_vs_reference') or {}).items() if k.startswith('pixels')})
PY
}
B="python bench.py --steps 60 --warmup 10 --no-other-configs --no-cpu-baseline --no-x6 --no-roofline --no-whole-clip"
for rep in 1 2; do
  echo "== two phases, pass $rep"; timeout 600 $B > $O/ab_on.json 2> $O/ab_on.err; one $O/ab_on.json
  echo "== round-5 step, pass $rep"; AOT_HIP_LIB=$L timeout 600 $B > $O/ab_old.json 2> $O/ab_old.err; one $O/ab_old.json
done
} > $O/r06_call15.txt 2>&1
cat $O/r06_call15.txt
