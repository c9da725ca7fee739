#!/bin/bash
# round 6: the evidence the driver's line leans on, re-measured on the library as built:
#  1. FETCH_SIZE / WRITE_SIZE PMC passes of the four flash attention kernels over the bench clip's launch mix -> r06_*traffic.json
#     (copied into profiles/ on the box first, so that the bench run below names them as its traffic source)
#  2. rocprofv3 --kernel-trace --stats of the bench command: R50-AOTL one clip and three clips, R50-DeAOTL and SwinB-DeAOTL one clip
#  3. the driver's command
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
pass() {  # mode, kernel substring, output json
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pm_$c
    timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/pm_$c -o p -- python $R/tools/dev/pmc_attn_mix.py $1 > $O/${TAG}_pm_$1_$c.log 2>&1 || echo "pass $1 $c failed"
  done
  python $R/tools/dev/attn_traffic.py $(find $O/pm_FETCH_SIZE -name "*.db" | head -1) $(find $O/pm_WRITE_SIZE -name "*.db" | head -1) $O/$3 $2 > /dev/null 2>&1
  python -c "import json; d=json.load(open('$O/$3')); print('$1', d['kernel'], 'launches', d['launches'], 'traffic/launch', round(d['traffic_bytes_per_launch']), d['bytes_per_launch'])"
  cp $O/$3 $R/profiles/$3
  rm -rf $O/pm_FETCH_SIZE $O/pm_WRITE_SIZE
}
pass aotx6 attn_x6_d32_kernel ${TAG}_attn_x6_traffic.json
pass aot attn_fwd_d32_pipe_kernel ${TAG}_attn_traffic.json
pass gatedx6 attn_x6_wide64p_kernel ${TAG}_gated_attn_x6_traffic.json
pass gated attn_fwd_wide_coop_kernel ${TAG}_gated_attn_traffic.json
prof() {  # model, streams, out name
  rm -rf $O/prof
  timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o t -- python $R/bench.py --model $1 --steps 207 --warmup 10 --no-other-configs --no-cpu-baseline --no-x6 --no-roofline --no-jf --no-whole-clip --streams $2 > $O/${TAG}_prof_$3.json 2> $O/${TAG}_prof_$3.err
  DB=$(find $O/prof -name "*.db" | head -1)
  python $R/tools/dev/prof_summary.py $DB $O/${TAG}_$3.txt | head -28 | cut -c1-125
  python $R/tools/dev/prof_timeline.py $DB $O/${TAG}_$3.txt
  rm -rf $O/prof
}
prof r50_aotl 1 bench_kernel_stats_s1
prof r50_aotl 3 bench_kernel_stats_s3
prof r50_deaotl 1 r50_deaotl_kernel_stats_s1
prof swinb_deaotl 1 swinb_deaotl_kernel_stats_s1
cd $R
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench20.json 2> $O/${TAG}_bench20.err
tail -1 $O/${TAG}_bench20.json | cut -c1-1200
grep -i "error\|fail\|Traceback" $O/${TAG}_bench20.err | head
