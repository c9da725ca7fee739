#!/bin/bash
# round 6, call 10: whole GPU suite on HEAD (after the 64-query gated kernel, the prune, the fp64 tie pins and the real-image fixture),
# then the one-clip kernel table of the bench command (the round's baseline for the launch-floor work)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 3300 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > $O/r06_suite_a.txt
cat $O/r06_suite_a.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o t -- python $R/bench.py --steps 207 --warmup 10 --no-other-configs --no-cpu-baseline --no-x6 --no-roofline --no-jf --no-whole-clip --streams 1 > $O/r06a_prof_s1.json 2> $O/r06a_prof_s1.err
DB=$(find $O/prof -name "*.db" | head -1)
python $R/tools/dev/prof_summary.py $DB $O/r06a_bench_kernel_stats_s1.txt | head -40 | cut -c1-125
python $R/tools/dev/prof_timeline.py $DB $O/r06a_bench_kernel_stats_s1.txt
rm -rf $O/prof
