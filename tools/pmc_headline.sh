#!/bin/bash
# SQ / LDS counters of the headline kernel (one rocprofv3 --pmc pass per counter set, kernel trace only): where the wavefronts' cycles go
# (parked at s_waitcnt, issue stalls, LDS array cycles and bank conflicts of the accumulator atomics, VALU).  usage (GPU box):
# tools/pmc_headline.sh <tag> [env assignments for the bench, e.g. PG_NO_PIPE=1]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-headline}; shift
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES"; do
  i=$((i+1))
  env "$@" timeout 300 rocprofv3 --pmc $set --kernel-trace -d $OUT/${TAG}_pmc$i -o x -- python $R/bench.py --no-cpu-baseline --no-traffic --no-variants --steps 3 --warmup 1 > $OUT/${TAG}_pmc$i.log 2>&1 < /dev/null
  python $R/tools/rocprof_summary.py $OUT/${TAG}_pmc$i/x_results.db | grep -A60 "^counters" | grep "pg_fast\|^counters" >> $OUT/${TAG}_pmc.txt
done
cat $OUT/${TAG}_pmc.txt
