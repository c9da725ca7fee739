#!/bin/bash
# round 5, call 16: the driver's command, then kernel stats of the timed legs
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05z_bench20.json 2> $O/r05z_bench20.err
tail -1 $O/r05z_bench20.json | cut -c1-600
grep -i "error\|fail\|Traceback" $O/r05z_bench20.err | head
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o t -- python $R/bench.py --steps 207 --warmup 10 --no-other-configs --no-cpu-baseline --no-x6 --no-roofline --no-jf --no-whole-clip --streams 1 > $O/r05z_prof_s1.json 2> $O/r05z_prof_s1.err
DB=$(find $O/prof -name "*.db" | head -1)
python $R/tools/dev/prof_summary.py $DB $O/r05z_bench_kernel_stats_s1.txt | head -34 | cut -c1-125
python $R/tools/dev/prof_timeline.py $DB $O/r05z_bench_kernel_stats_s1.txt
rm -rf $O/prof
