#!/bin/bash
# round 5, call d: what bounds the partition scatter?  SQ + memory counters of the 40 k-group row, quad phase A vs oct phase A (3 workgroups per CU)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
for knob in "PG_NO_P2_OCT=1" "PG_P2_WGS_PER_CU=3"; do
  echo "# knob: $knob" >> $OUT/r05_d_counters.txt
  env $knob PMC_MEMORY=1 timeout 500 python tools/pmc_sq.py general "=group g1,g2,c_inv1 (40k groups)" 100000000 2>&1 | grep -v amdgpu.ids >> $OUT/r05_d_counters.txt
done
cat $OUT/r05_d_counters.txt
