#!/bin/bash
# x6 d32 attention at four waves per SIMD (register cap 128, the spills outside the loop); key-split sweep; encoder look-ahead sweep
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
L=aot-benchmark_amd/csrc
python tools/dev/mb_attn_x6.py "" quick > /dev/null 2>&1      # warm-up, discarded
{
for v in "" _lb4 "" _lb4; do
  echo "== attention, lib libaot_hip$v.so"
  timeout 200 python tools/dev/mb_attn_x6.py $L/libaot_hip$v.so quick 2>&1 | grep -v amdgpu.ids
done
for v in "" _lb4; do
  echo "== attention key-split sweep, lib libaot_hip$v.so"
  timeout 300 python tools/dev/mb_attn_x6.py $L/libaot_hip$v.so sweep 2>&1 | grep "^M="
done
} > $O/r04_x6_occupancy.txt 2>&1
B="--gpus 1 --steps 207 --warmup 5 --no-other-configs --no-cpu-baseline --no-x6 --no-jf --no-roofline --repeats 2"
{
for a in 3 5 7; do
  timeout 300 python bench.py $B --encode-ahead $a 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ahead $a fps', d['value'], d['config']['repeat_fps'], 'single', d['config']['single_stream']['fps'])"
done
AOT_HIP_LIB=$PWD/$L/libaot_hip_lb4.so timeout 300 python bench.py $B --encode-ahead 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lb4 ahead 3 fps', d['value'], d['config']['repeat_fps'], 'single', d['config']['single_stream']['fps'])"
} > $O/r04_ahead_sweep.txt 2>&1
cat $O/r04_x6_occupancy.txt | cut -c1-150 | head -60; cat $O/r04_ahead_sweep.txt
