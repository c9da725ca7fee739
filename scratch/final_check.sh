R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 100 python scratch/host_issue.py 2>&1 | grep "rep 2"
timeout 250 python bench.py 2>&1 | tail -1 > $O/bench_final.json; cat $O/bench_final.json
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_f
timeout 250 rocprofv3 --kernel-trace --stats -d $O/prof_f -o p -- python $R/bench.py --no-cpu-baseline > $O/prof_f_bench.log 2>&1
cd $R
python scratch/prof_summary.py $(find $O/prof_f -name "*.db" | head -1) $O/bench_kernel_stats.txt | head -16
rm -rf $O/prof_f
