#!/bin/bash
# round 6, call 14: softmax weights on two planes (AOT_P16 build of attn_x6_d32_kernel): launch time and error against fp64, the kernel's
# unit tests, the free-running parity cells of R50-AOTL, alternating bench runs
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
L=$R/aot-benchmark_amd/csrc/libaot_hip_p16.so
{
echo "== product"; timeout 300 python tools/dev/mb_attn_x6.py "" 2>&1 | grep -v amdgpu.ids
echo "== AOT_P16"; timeout 300 python tools/dev/mb_attn_x6.py $L 2>&1 | grep -v amdgpu.ids
echo "== gated, product"; timeout 300 python tools/dev/mb_gated_x6.py "" quick 2>&1 | grep -v amdgpu.ids
echo "== gated, AOT_P16"; timeout 300 python tools/dev/mb_gated_x6.py $L quick 2>&1 | grep -v amdgpu.ids
echo "== unit tests, AOT_P16 library"
AOT_HIP_LIB=$L timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -k "attention_x6 or attention_kernels_reproducible or gated_attention_x6" 2>&1 | tail -12
echo "== parity cells of R50-AOTL, AOT_P16 library"
AOT_HIP_LIB=$L timeout 2400 python -m pytest tests/test_parity_gpu.py -q -m gpu -k "bf16x6 and (c2_r50_aotl_70 or c3b_r50_deaotl_70)" 2>&1 | tail -12
mv $O/parity_r06.json $O/parity_p16.json
python - <<'PY'
import json
d = json.load(open('gpurun_out/parity_p16.json'))
for k, v in sorted(d.items()):
    print(k, json.dumps({kk: v[kk] for kk in v if kk in ('pixels_differing', 'differing', 'outside', 'pixels_outside_near_ties', 'max_logit_err', 'logits_max_abs_err', 'on_fp64', 'per_frame_max')})[:400])
PY
one() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d['config']
print(' value', d['value'], c.get('repeat_fps'), 'single', (c.get('single_stream') or {}).get('fps'), 'jf', c.get('jf_vs_reference'))
PY
}
B="python bench.py --steps 60 --warmup 10 --no-other-configs --no-cpu-baseline --no-x6 --no-roofline --no-whole-clip"
echo "== swinb_deaotl product"; timeout 600 $B --model swinb_deaotl > $O/ab_on.json 2> $O/ab_on.err; one $O/ab_on.json
echo "== swinb_deaotl AOT_P16"; AOT_HIP_LIB=$L timeout 600 $B --model swinb_deaotl > $O/ab_p16.json 2> $O/ab_p16.err; one $O/ab_p16.json
for rep in 1 2; do
  echo "== product, pass $rep"; timeout 600 $B > $O/ab_on.json 2> $O/ab_on.err; one $O/ab_on.json
  echo "== AOT_P16, pass $rep"; AOT_HIP_LIB=$L timeout 600 $B > $O/ab_p16.json 2> $O/ab_p16.err; one $O/ab_p16.json
done
} > $O/r06_call14.txt 2>&1
cat $O/r06_call14.txt
