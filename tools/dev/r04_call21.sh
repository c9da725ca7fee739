#!/bin/bash
# Is the x6 d32 attention kernel bound by its K / V fetches?  Probe build: every key tile reads the bank's first tile (L1 hits).
# + encoder look-ahead sweep beyond 7
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
L=aot-benchmark_amd/csrc
python tools/dev/mb_attn_x6.py "" quick > /dev/null 2>&1      # warm-up, discarded
{
for v in "" _sametile "" _sametile; do
  echo "== attention, lib libaot_hip$v.so"
  timeout 200 python tools/dev/mb_attn_x6.py $L/libaot_hip$v.so quick 2>&1 | grep "^M=" | cut -c1-100
done
} > $O/r04_x6_sametile_probe.txt 2>&1
B="--gpus 1 --steps 207 --warmup 5 --no-other-configs --no-cpu-baseline --no-x6 --no-jf --no-roofline --repeats 2"
{
for a in 7 10 14 23; do
  timeout 300 python bench.py $B --encode-ahead $a 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ahead $a fps', d['value'], d['config']['repeat_fps'], 'single', d['config']['single_stream']['fps'], 'mem', d['config'].get('peak_mem_gib'))"
done
} > $O/r04_ahead_sweep2.txt 2>&1
cat $O/r04_x6_sametile_probe.txt $O/r04_ahead_sweep2.txt
