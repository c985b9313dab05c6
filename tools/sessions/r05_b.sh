#!/bin/bash
# round 5, call b: the oct-layout partition scatter — parity tests, then the high-cardinality rows with and without it, then a kernel trace
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_gpu_partition_pipeline.py tests/test_gpu_oct.py tests/test_raw_group_by.py -x -q -m gpu 2>&1 | tail -8 > $OUT/r05_b_tests.txt
cat $OUT/r05_b_tests.txt
for knob in "" "PG_NO_P2_OCT=1"; do
  echo "# knob: ${knob:-default}" >> $OUT/r05_b_variants.txt
  env $knob timeout 200 python tools/prof_variants.py --set general --docs 200000000 --only 0k 2>&1 < /dev/null | grep -v amdgpu.ids >> $OUT/r05_b_variants.txt
  env $knob timeout 200 python tools/prof_variants.py --set cfg5 --docs 200000000 --only "1M groups" 2>&1 < /dev/null | grep -v amdgpu.ids >> $OUT/r05_b_variants.txt
done
cat $OUT/r05_b_variants.txt | cut -c1-160
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace -d $OUT/r05_b_prof -o x -- python $R/tools/prof_variants.py --set general --docs 200000000 --only 0k > /dev/null 2>&1 < /dev/null )
timeout 60 python tools/rocprof_summary.py $OUT/r05_b_prof/x_results.db > $OUT/r05_b_kernel_stats.txt 2>&1
head -30 $OUT/r05_b_kernel_stats.txt | cut -c1-150
