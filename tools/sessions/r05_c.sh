#!/bin/bash
# round 5, call c: rank atomics never under a per-lane branch (all scatter kernels), oct phase A at 4 and 3 workgroups per CU, 3-deep aggregate
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_gpu_partition_pipeline.py tests/test_gpu_oct.py -x -q -m gpu 2>&1 | tail -5 > $OUT/r05_c_tests.txt
cat $OUT/r05_c_tests.txt
for knob in "PG_X=1" "PG_P2_WGS_PER_CU=3" "PG_NO_P2_OCT=1"; do
  echo "# knob: ${knob}" >> $OUT/r05_c_variants.txt
  env $knob timeout 200 python tools/prof_variants.py --set general --docs 200000000 --only 0k 2>&1 < /dev/null | grep -v amdgpu.ids >> $OUT/r05_c_variants.txt
  env $knob timeout 200 python tools/prof_variants.py --set cfg5 --docs 200000000 --only "1M groups" 2>&1 < /dev/null | grep -v amdgpu.ids >> $OUT/r05_c_variants.txt
done
cat $OUT/r05_c_variants.txt | cut -c1-160
for knob in "PG_X=1" "PG_P2_WGS_PER_CU=3"; do
( cd /tmp && export TMPDIR=/tmp && env $knob timeout 200 rocprofv3 --kernel-trace -d $OUT/r05_c_prof_$knob -o x -- python $R/tools/prof_variants.py --set general --docs 200000000 --only 0k > /dev/null 2>&1 < /dev/null )
timeout 60 python tools/rocprof_summary.py $OUT/r05_c_prof_$knob/x_results.db > $OUT/r05_c_kernel_stats_$knob.txt 2>&1
head -12 $OUT/r05_c_kernel_stats_$knob.txt | cut -c1-150
done
