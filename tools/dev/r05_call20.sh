#!/bin/bash
# round 5, call 20: the ResNet stem in the bf16x6 family (aot_conv2d_c4_bf16x6_f32): kernel test, parity cells, bench A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
{
timeout 600 python -m pytest tests/test_parity_gpu.py -q -m gpu -k "c4_bf16x6 or (bf16x6 and tail and r50) or graph_replay_bit_identical or encode_ahead" 2>&1 | tail -5
F="--steps 207 --warmup 10 --no-other-configs --no-cpu-baseline --no-x6 --no-roofline --no-whole-clip"
for e in 1 "" 1 ""; do
  echo "== bench, AOT_NO_C4=${e:-0}"; AOT_NO_C4=$e timeout 600 python bench.py $F $( [ -n "$e" ] && echo --no-jf ) 2>/dev/null | tail -1 > $O/_b.json
  python - <<'PY'
import json
d = json.loads(open('gpurun_out/_b.json').read().strip().splitlines()[-1]); c = d['config']
print(' value', d['value'], c.get('repeat_fps'), 'single', (c.get('single_stream') or {}).get('fps'),
      'jf', {k: (c.get('jf_vs_reference') or {}).get(k) for k in ('pixels_differing', 'pixels_outside_near_ties')})
PY
done
} > $O/r05_stem_c4.txt 2>&1
cat $O/r05_stem_c4.txt
