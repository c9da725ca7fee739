#!/bin/bash
# round 5, call 5: kernel stats + concurrency of the bench's timed legs with the round-5 GEMM dispatch (three clips / one clip)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
F="--steps 207 --warmup 10 --no-other-configs --no-cpu-baseline --no-x6 --no-roofline --no-jf --no-whole-clip"
for s in 3 1; do
  rm -rf $O/prof
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o t -- python $R/bench.py $F --streams $s > $O/r05_prof_s$s.json 2> $O/r05_prof_s$s.err
  DB=$(find $O/prof -name "*.db" | head -1)
  python $R/tools/dev/prof_summary.py $DB $O/r05_bench_kernel_stats_s$s.txt | head -45 | cut -c1-125
  python $R/tools/dev/prof_timeline.py $DB $O/r05_bench_kernel_stats_s$s.txt
  tail -1 $O/r05_prof_s$s.json | cut -c1-300
done
rm -rf $O/prof
