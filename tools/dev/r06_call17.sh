#!/bin/bash
# round 6, call 17: LayerNorm-prologue GEMMs in the Swin-B trunk (norm1 -> qkv, norm2 -> fc1, patch-merging norm -> reduction): Swin tests,
# the config-3 goldens, A/B of AOT_NO_LN_FUSE on SwinB-DeAOTL
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
{
timeout 1200 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "swin or layernorm_linear" 2>&1 | tail -5
timeout 2400 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "bf16x6 and c3_swinb_deaotl_480" 2>&1 | tail -5
one() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d['config']
print(' value', d['value'], c.get('repeat_fps'), 'single', (c.get('single_stream') or {}).get('fps'), 'jf', {k: v for k, v in (c.get('jf_vs_reference') or {}).items() if k.startswith('pixels')})
PY
}
B="python bench.py --model swinb_deaotl --steps 60 --warmup 10 --no-other-configs --no-cpu-baseline --no-x6 --no-roofline --no-whole-clip"
for rep in 1 2; do
  echo "== swinb_deaotl fused (default), pass $rep"; timeout 900 $B > $O/ab_on.json 2> $O/ab_on.err; one $O/ab_on.json
  echo "== swinb_deaotl AOT_NO_LN_FUSE, pass $rep"; AOT_NO_LN_FUSE=1 timeout 900 $B > $O/ab_off.json 2> $O/ab_off.err; one $O/ab_off.json
done
} > $O/r06_call17.txt 2>&1
cat $O/r06_call17.txt
