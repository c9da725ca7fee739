#!/bin/bash
# round 5, call 4: the new bf16x6 GEMM dispatch (register-staged 64x64 by default, 128x128 for the big KxK layers, split-K phase-shifted
# kernel for the long-K stride-16 layers) against the round-4 rule (AOT_X6_TILE=1) in the bench itself
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
F="--steps 207 --warmup 10 --no-other-configs --no-cpu-baseline --no-x6 --no-roofline"
{
echo "== round-4 dispatch"; AOT_X6_TILE=1 timeout 600 python bench.py $F --no-jf 2>/dev/null | tail -1
echo "== round-5 dispatch"; timeout 600 python bench.py $F 2>/dev/null | tail -1
echo "== round-4 dispatch, r50_deaotl"; AOT_X6_TILE=1 timeout 600 python bench.py $F --no-jf --model r50_deaotl 2>/dev/null | tail -1
echo "== round-5 dispatch, r50_deaotl"; timeout 600 python bench.py $F --model r50_deaotl 2>/dev/null | tail -1
echo "== round-5 dispatch, swinb_deaotl"; timeout 600 python bench.py $F --model swinb_deaotl 2>/dev/null | tail -1
} > $O/r05_bench_dispatch.txt 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/r05_bench_dispatch.txt'):
    if l.startswith('=='): print(l.strip()); continue
    try: d = json.loads(l)
    except Exception: print(l[:300]); continue
    c = d['config']
    print(' value', d['value'], c.get('repeat_fps'), 'single', c.get('single_stream', {}).get('fps'), 'whole', c.get('whole_clip', {}).get('fps'),
          'jf', {k: c.get('jf_vs_reference', {}).get(k) for k in ('pixels_differing', 'pixels_outside_near_ties')})
PY
