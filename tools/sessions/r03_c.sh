#!/bin/bash
# round 3, GPU call C: parity of the reworked partition pipeline + timings + kernel breakdown
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O; TAG=${1:-r03_c}
timeout 420 python -m pytest tests/test_gpu_partition_pipeline.py -q > $O/${TAG}_tests_part.log 2>&1; echo "part tests rc=$?"
tail -8 $O/${TAG}_tests_part.log
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "radix or partitioned or config5 or hashed or num_groups or distinct" > $O/${TAG}_tests_parity.log 2>&1; echo "parity subset rc=$?"
tail -3 $O/${TAG}_tests_parity.log
for set in cfg5 general; do
  timeout 200 python tools/prof_variants.py --set $set --docs 200000000 --reps 6 > $O/${TAG}_${set}_new.txt 2>&1
  echo "== $set new"; cat $O/${TAG}_${set}_new.txt | grep -v "^#" | grep "part_group\|radix_group"
done
cd /tmp && export TMPDIR=/tmp
for set in cfg5 general; do
  only="cfg5"; [ $set = general ] && only="group"
  timeout 200 rocprofv3 --kernel-trace -d $O/${TAG}_${set}_trace -o x -- python $R/tools/prof_variants.py --set $set --docs 200000000 --reps 4 --only "$only" > $O/${TAG}_${set}_trace.log 2>&1 < /dev/null
  python $R/tools/rocprof_summary.py $O/${TAG}_${set}_trace/x_results.db > $O/${TAG}_${set}_kernel_stats.txt 2>&1
  grep -E "^kernel|pg_p2|pg_radix" $O/${TAG}_${set}_kernel_stats.txt | cut -c1-160
  rm -rf $O/${TAG}_${set}_trace
done
