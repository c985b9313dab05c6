#!/bin/bash
# round 3, GPU call B: kernel-level breakdown of the partition pipeline (rocprofv3 kernel trace)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for set in cfg5 general; do
  only="cfg5"; [ $set = general ] && only="group"
  timeout 200 rocprofv3 --kernel-trace -d $O/r03_b_${set}_trace -o x -- python $R/tools/prof_variants.py --set $set --docs 200000000 --reps 4 --only "$only" > $O/r03_b_${set}_trace.log 2>&1 < /dev/null
  python $R/tools/rocprof_summary.py $O/r03_b_${set}_trace/x_results.db > $O/r03_b_${set}_kernel_stats.txt 2>&1
  head -20 $O/r03_b_${set}_kernel_stats.txt | cut -c1-160
  rm -rf $O/r03_b_${set}_trace
done
