cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
for m in swinb_deaotl r50_deaotl; do
  rm -rf $O/prof_$m
  timeout 80 rocprofv3 --kernel-trace --stats -d $O/prof_$m -o p -- python $R/scratch/dev_time2.py $m 70 > $O/prof_$m.log 2>&1
  grep fps $O/prof_$m.log
  (cd $R && python scratch/prof_summary.py $(find $O/prof_$m -name "*.db" | head -1) $O/${m}_kernel_stats.txt | head -14)
  rm -rf $O/prof_$m
done
