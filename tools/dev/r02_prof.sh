# round-2 GPU check: tests, bench (driver form), rocprof kernel stats of the bench
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc=$?" 
tail -3 $O/gputest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench20.log 2>&1; tail -1 $O/bench20.log
cd /tmp
rm -rf $O/prof_f
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_f -o p -- python $R/bench.py --no-cpu-baseline > $O/prof_f_bench.log 2>&1
cd $R
python tools/dev/prof_summary.py $(find $O/prof_f -name "*.db" | head -1) $O/bench_kernel_stats.txt | head -40
tail -1 $O/prof_f_bench.log
rm -rf $O/prof_f
