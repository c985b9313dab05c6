#!/bin/bash
# per-kernel split of the high-cardinality GROUP BYs (partition pipeline) under rocprofv3 --kernel-trace: usage (GPU box) tools/trace_high_cardinality.sh ["only" ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ $# -eq 0 ] && set -- "group g1,g2,c_inv1 (40k" "group g1,g2,c_inv1,c_inv2 (160k)" "count group u (1M groups)"
for only in "$@"; do
  rm -rf /tmp/tr; timeout 200 rocprofv3 --kernel-trace -d /tmp/tr -o x -- python $R/tools/prof_variants.py --set general --docs 200000000 --reps 5 --only "$only" > /tmp/tr.log 2>&1 < /dev/null
  grep -E "ms " /tmp/tr.log | head -3
  python $R/tools/rocprof_summary.py /tmp/tr/x_results.db 2>/dev/null | head -8 | cut -c1-130
done
