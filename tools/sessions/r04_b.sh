#!/bin/bash
# round 4, session b: oct kernels after the per-pass launch diet; where the pruned passes spend their time (kernel trace), SQ counters of pg_oct_l
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O; TAG=${1:-r04_b}
timeout 600 python -m pytest tests/test_gpu_oct.py tests/test_gpu_bench_contract.py tests/test_mv_reference_goldens.py tests/test_gpu_multi.py tests/test_gpu_partition_pipeline.py -q -p no:cacheprovider > $O/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -25 $O/${TAG}_tests.log | cut -c1-300
timeout 200 python tools/prof_variants.py --set cfg5 --docs 200000000 --reps 8 > $O/${TAG}_cfg5_oct.txt 2>&1; grep -v "^/opt" $O/${TAG}_cfg5_oct.txt
cd /tmp; export TMPDIR=/tmp
for q in "=cfg5" "=cfg5 hll(u) group h1"; do
  n=$(echo "$q" | tr -c 'a-z0-9' '_')
  rm -rf /tmp/prof_$n; timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_$n -o x -- python $R/tools/prof_variants.py --set cfg5 --only "$q" --docs 200000000 --reps 6 > $O/${TAG}_prof_$n.log 2>&1
  db=$(find /tmp/prof_$n -name "*_results.db" | head -1); python $R/tools/rocprof_summary.py $db > $O/${TAG}_kernels_$n.txt 2>&1; echo "== $q"; head -30 $O/${TAG}_kernels_$n.txt
done
cd $R
timeout 500 python tools/pmc_sq.py cfg5 "=cfg5 hll(u) group h1" 200000000 > $O/${TAG}_sq_oct_l.txt 2>&1; grep -v "^/opt" $O/${TAG}_sq_oct_l.txt | head -60
timeout 500 python tools/pmc_sq.py cfg5 "=cfg5" 200000000 > $O/${TAG}_sq_cfg5.txt 2>&1; grep -v "^/opt" $O/${TAG}_sq_cfg5.txt | head -150
