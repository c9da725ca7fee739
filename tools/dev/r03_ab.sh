# A/B of prebuilt library variants inside ONE gpurun call (first process on a fresh box reads ~10 % low: a throw-away warm-up
# run first, then every variant under the same conditions).  Usage on the box:  bash tools/dev/r03_ab.sh base epi1 pksum1 both
# where `base` is the product library and the others were built here with tools/dev/build_variant.sh.
cd $GRAFT_REPO_ROOT; O=gpurun_out; L=aot-benchmark_amd/csrc
cp $L/libaot_hip.so /tmp/libaot_hip_base.so
timeout 200 python bench.py --no-cpu-baseline --no-jf --no-roofline > /dev/null 2>&1      # warm-up, discarded
for v in "$@"; do
  if [ "$v" = base ]; then cp /tmp/libaot_hip_base.so $L/libaot_hip.so; else cp $L/libaot_hip_$v.so $L/libaot_hip.so; fi
  timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -p no:cacheprovider \
      -k "conv2d or linear or attention or end_to_end_vs or free_running_masks" > $O/ab_$v.test.log 2>&1
  echo "$v: pytest rc=$? $(tail -1 $O/ab_$v.test.log)"
  timeout 300 python bench.py --no-cpu-baseline --no-jf > $O/ab_$v.bench.json 2> $O/ab_$v.bench.err
  python -c "import json,sys; d=json.load(open('$O/ab_$v.bench.json')); print('$v', d['value'], d['config']['single_stream']['fps'], d['roofline']['achieved'])"
  case $v in base|*coop*|all*) timeout 200 python bench.py --model r50_deaotl --no-cpu-baseline > $O/ab_$v.deaot.json 2>/dev/null
    python -c "import json; d=json.load(open('$O/ab_$v.deaot.json')); print('$v r50_deaotl', d['value'], d['config']['single_stream']['fps'])";; esac
  timeout 300 python tools/dev/mb_gemm.py -1,197 > $O/ab_$v.gemm.txt 2>&1; tail -2 $O/ab_$v.gemm.txt       # (uses the library in place)
done
cp /tmp/libaot_hip_base.so $L/libaot_hip.so
