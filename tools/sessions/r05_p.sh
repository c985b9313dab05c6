#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_gpu_partition_pipeline.py -x -q -m gpu 2>&1 | tail -3
for knob in "PG_X=1" "PG_P2_AGG_WGS_PER_CU=1"; do
  echo "# knob: ${knob}" >> $OUT/r05_p_variants.txt
  env $knob timeout 200 python tools/prof_variants.py --set general --docs 200000000 --only 0k 2>&1 < /dev/null | grep -v amdgpu.ids >> $OUT/r05_p_variants.txt
  env $knob timeout 200 python tools/prof_variants.py --set cfg5 --docs 200000000 --only "1M groups" 2>&1 < /dev/null | grep -v amdgpu.ids >> $OUT/r05_p_variants.txt
done
cat $OUT/r05_p_variants.txt | cut -c1-160
