#!/bin/bash
# round 5, call e: cycles per phase of the partition scatter's round (s_memtime variant)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
for knob in "PG_NO_P2_OCT=1" "PG_P2_WGS_PER_CU=3" "PG_X=1"; do
  echo "# knob: $knob" >> $OUT/r05_e_phases.txt
  env $knob PG_GPU_LIB=$R/tools/variants/libpinot_gpu_p2timing.so timeout 300 python tools/prof_variants.py --set general --docs 100000000 --reps 2 --only "0k" 2>&1 | grep -v amdgpu.ids | awk '!seen[$0]++' >> $OUT/r05_e_phases.txt
  env $knob PG_GPU_LIB=$R/tools/variants/libpinot_gpu_p2timing.so timeout 300 python tools/prof_variants.py --set cfg5 --docs 100000000 --reps 2 --only "1M groups" 2>&1 | grep -v amdgpu.ids | awk '!seen[$0]++' >> $OUT/r05_e_phases.txt
done
cat $OUT/r05_e_phases.txt | cut -c1-250
