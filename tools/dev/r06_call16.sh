#!/bin/bash
# round 6, call 16: (i) forked attention branches + overlapped look-ahead: unit test, one-clip A/B; (ii) the linears of the Swin-B trunk on the
# 64x64 direct-weight kernel against the register-staged 128x128 tile, batch 3 (what the look-ahead encoder runs) and batch 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
{
timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "overlapped_encode_ahead or encode_ahead_matches" 2>&1 | tail -4
one() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d['config']
print(' value', d['value'], 'single', (c.get('single_stream') or {}).get('fps'), (c.get('single_stream') or {}).get('repeat_fps'), {k: v for k, v in (c.get('single_stream') or {}).items() if 'forked' in k or 'overl' in k})
PY
}
B="python bench.py --steps 60 --warmup 10 --no-other-configs --no-cpu-baseline --no-x6 --no-roofline --no-jf --no-whole-clip"
for rep in 1 2; do
  echo "== branches + overlap (default), pass $rep"; timeout 600 $B > $O/ab_on.json 2> $O/ab_on.err; one $O/ab_on.json
  echo "== AOT_NO_BRANCHES, pass $rep"; AOT_NO_BRANCHES=1 timeout 600 $B > $O/ab_off.json 2> $O/ab_off.err; one $O/ab_off.json
done
echo "== Swin-B linears, batch 3: dispatch / 64x64 direct-weight / 128x128 register-staged"
AOT_MB_SHAPES=swin timeout 600 python tools/dev/mb_gemm.py x6,x6d,x6s "" "" 3 2>&1 | grep -v amdgpu.ids
echo "== Swin-B linears, batch 1"
AOT_MB_SHAPES=swin timeout 600 python tools/dev/mb_gemm.py x6,x6d,x6s "" "" 1 2>&1 | grep -v amdgpu.ids
echo "== R50 frame set, batch 3: 1x1 layers on the 128x128 tile"
timeout 600 python tools/dev/mb_gemm.py x6,x6s "" "" 3 2>&1 | grep -v amdgpu.ids
} > $O/r06_call16.txt 2>&1
cat $O/r06_call16.txt
