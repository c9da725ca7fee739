#!/bin/bash
# round 5, call 19: GroupNorm partial sums out of the GEMM tile end (LSTT feed-forward): kernel test, parity cells, bench A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
{
timeout 600 python -m pytest tests/test_parity_gpu.py -q -m gpu -k "gn_partials or (bf16x6 and tail and c2_r50_aotl_70) or (free_running and c1_aott) or graph_replay_bit_identical" 2>&1 | tail -5
F="--steps 207 --warmup 10 --no-other-configs --no-cpu-baseline --no-x6 --no-roofline --no-whole-clip"
for e in 1 "" 1 ""; do
  echo "== bench, AOT_NO_GN_FUSE=${e:-0}"; AOT_NO_GN_FUSE=$e timeout 600 python bench.py $F $( [ -n "$e" ] && echo --no-jf ) 2>/dev/null | tail -1 > $O/_b.json
  python - <<'PY'
import json
d = json.loads(open('gpurun_out/_b.json').read().strip().splitlines()[-1]); c = d['config']
print(' value', d['value'], c.get('repeat_fps'), 'single', (c.get('single_stream') or {}).get('fps'),
      'jf', {k: (c.get('jf_vs_reference') or {}).get(k) for k in ('pixels_differing', 'pixels_outside_near_ties')})
PY
done
} > $O/r05_gnfuse.txt 2>&1
cat $O/r05_gnfuse.txt
