#!/bin/bash
# round 6, call 4: pipelined 64-query gated kernel in the engine: DeAOT parity cells, DeAOT benches (XCD-major vs linear order)
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
{
timeout 1500 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "gated or (deaot and bf16x6)" 2>&1 | tail -6
F="--steps 20 --warmup 5 --no-other-configs --no-cpu-baseline --no-x6 --no-whole-clip"
for m in r50_deaotl swinb_deaotl; do
  for lib in "" aot-benchmark_amd/csrc/libaot_hip_plin.so; do
    echo "== bench $m lib=${lib:-product}"; AOT_HIP_LIB=$lib timeout 900 python bench.py $F --model $m 2>/dev/null | tail -1 > $O/_b.json
    python - <<'PY'
import json
d = json.loads(open('gpurun_out/_b.json').read().strip().splitlines()[-1]); c = d['config']
print(' value', d['value'], c.get('repeat_fps'), 'single', (c.get('single_stream') or {}).get('fps'),
      'jf', {k: (c.get('jf_vs_reference') or {}).get(k) for k in ('pixels_differing', 'pixels_outside_near_ties')},
      'roofline', {k: d.get('roofline', {}).get(k) for k in ('kernel', 'frac', 'achieved', 'avg_launch_us')})
PY
  done
done
} > $O/r06_gated64_engine.txt 2>&1
cat $O/r06_gated64_engine.txt
