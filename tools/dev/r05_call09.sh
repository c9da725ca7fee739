#!/bin/bash
# round 5, call 9: adapters with the encoder, vectorised split-K reduce, staging copy skipped: parity cells of the timed configuration,
# bench, encoder look-ahead sweep
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
{
timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -k "frame_tail or phase_shifted or (bf16x6 and tail) or encode_ahead or (free_running and c2_r50_aotl_70 and f32 and throughput)" 2>&1 | tail -5
F="--steps 207 --warmup 10 --no-other-configs --no-cpu-baseline --no-x6 --no-roofline --no-whole-clip"
for a in 3 7 23; do
  echo "== bench, encode-ahead $a"; timeout 600 python bench.py $F --encode-ahead $a $( [ $a != 3 ] && echo --no-jf ) 2>/dev/null | tail -1 > $O/_b.json
  python - <<'PY'
import json
d = json.loads(open('gpurun_out/_b.json').read()); c = d['config']
print(' value', d['value'], c.get('repeat_fps'), 'single', (c.get('single_stream') or {}).get('fps'), 'mem', c.get('peak_mem_gib'),
      'jf', {k: (c.get('jf_vs_reference') or {}).get(k) for k in ('pixels_differing', 'pixels_outside_near_ties')})
PY
done
} > $O/r05_call09.txt 2>&1
cat $O/r05_call09.txt
