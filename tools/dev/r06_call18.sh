#!/bin/bash
# round 6, call 18: Swin window attention as ONE launch per block for the whole look-ahead batch (image = grid z) against one launch per image
# (AOT_SWIN_PER_IMAGE); Swin tests + the config-3 goldens; clip streams 2 / 4 (with the overlapped look-ahead) against 3
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
{
timeout 1200 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "swin" 2>&1 | tail -5
timeout 2400 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "bf16x6 and c3_swinb_deaotl_480" 2>&1 | tail -5
one() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d['config']
print(' value', d['value'], c.get('repeat_fps'), 'single', (c.get('single_stream') or {}).get('fps'), 'jf', {k: v for k, v in (c.get('jf_vs_reference') or {}).items() if k.startswith('pixels')})
PY
}
B="python bench.py --model swinb_deaotl --steps 60 --warmup 10 --no-other-configs --no-cpu-baseline --no-x6 --no-roofline --no-whole-clip"
for rep in 1 2; do
  echo "== swinb_deaotl batched window attention (default), pass $rep"; timeout 900 $B > $O/ab_on.json 2> $O/ab_on.err; one $O/ab_on.json
  echo "== swinb_deaotl AOT_SWIN_PER_IMAGE, pass $rep"; AOT_SWIN_PER_IMAGE=1 timeout 900 $B > $O/ab_off.json 2> $O/ab_off.err; one $O/ab_off.json
done
B="python bench.py --steps 60 --warmup 10 --no-other-configs --no-cpu-baseline --no-x6 --no-roofline --no-jf --no-whole-clip"
echo "== r50_aotl streams 3 (default)"; timeout 600 $B > $O/ab_x.json 2> $O/ab_x.err; one $O/ab_x.json
echo "== r50_aotl streams 2 + overlap"; timeout 600 $B --streams 2 --overlap-encode 1 > $O/ab_x.json 2> $O/ab_x.err; one $O/ab_x.json
echo "== r50_aotl streams 2"; timeout 600 $B --streams 2 --overlap-encode 0 > $O/ab_x.json 2> $O/ab_x.err; one $O/ab_x.json
echo "== r50_aotl streams 4"; timeout 600 $B --streams 4 > $O/ab_x.json 2> $O/ab_x.err; one $O/ab_x.json
} > $O/r06_call18.txt 2>&1
cat $O/r06_call18.txt
