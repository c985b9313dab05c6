#!/bin/bash
# round 5, call h: numGroupsLimit by a prefix pass — parity tests, then the rows it moves
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_partition_pipeline.py tests/test_raw_group_by.py tests/test_fuzz.py -x -q -m gpu 2>&1 | tail -8 > $OUT/r05_h_tests.txt
cat $OUT/r05_h_tests.txt
for knob in "PG_X=1" "PG_NO_LIMIT_PREFIX=1"; do
  echo "# knob: ${knob}" >> $OUT/r05_h_variants.txt
  env $knob timeout 200 python tools/prof_variants.py --set general --docs 200000000 --only "160k" 2>&1 < /dev/null | grep -v amdgpu.ids >> $OUT/r05_h_variants.txt
  env $knob timeout 200 python tools/prof_variants.py --set cfg5 --docs 200000000 --only "1M groups" 2>&1 < /dev/null | grep -v amdgpu.ids >> $OUT/r05_h_variants.txt
done
cat $OUT/r05_h_variants.txt | cut -c1-160
