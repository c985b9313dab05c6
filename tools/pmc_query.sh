#!/bin/bash
# rocprofv3 PMC passes (counters in their own runs, kernel-trace only) over tools/prof_variants.py for one query set.
# usage (on the GPU box): tools/pmc_query.sh <set> <docs> <grep-pattern-of-query-name>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY" "FETCH_SIZE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  n=$(echo $set | cut -d" " -f1)
  timeout 280 rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmcq_$n -o x -- python $R/tools/prof_variants.py --set $1 --docs $2 --reps 2 --only "$3" > $R/gpurun_out/pmcq_$n.log 2>&1
  python $R/tools/rocprof_summary.py $R/gpurun_out/pmcq_$n/x_results.db | grep -A12 "^counters" | grep -v "^counters"
done
