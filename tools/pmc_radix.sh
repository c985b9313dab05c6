#!/bin/bash
# Kernel trace + HBM-traffic PMC passes (FETCH_SIZE and WRITE_SIZE in their own runs, kernel-trace only — MI355X_MICROARCH.md
# §rocprofv3 PMC slots) over the config-5 shapes of tools/prof_variants.py.  usage (GPU box): tools/pmc_radix.sh <tag> [docs] [only]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-radix}; DOCS=${2:-200000000}; ONLY=${3:-cfg5}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace -d $OUT/${TAG}_trace -o x -- python $R/tools/prof_variants.py --set cfg5 --docs $DOCS --reps 4 --only "$ONLY" > $OUT/${TAG}_trace.log 2>&1 < /dev/null
python $R/tools/rocprof_summary.py $OUT/${TAG}_trace/x_results.db > $OUT/${TAG}_kernel_stats.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace -d $OUT/${TAG}_pmc_$c -o x -- python $R/tools/prof_variants.py --set cfg5 --docs $DOCS --reps 2 --only "$ONLY" > $OUT/${TAG}_pmc_$c.log 2>&1 < /dev/null
  python $R/tools/rocprof_summary.py $OUT/${TAG}_pmc_$c/x_results.db | grep -A40 "^counters" | grep -v "^counters" >> $OUT/${TAG}_pmc.txt
done
head -14 $OUT/${TAG}_kernel_stats.txt | cut -c1-150
cat $OUT/${TAG}_pmc.txt
