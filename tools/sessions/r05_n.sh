#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
for knob in "PG_X=1" "PG_NO_FUSED_FINISH=1"; do
  echo "# $knob" >> $OUT/r05_n_startree.txt
  env $knob PG_TRACE_HOST=1 timeout 200 python tools/startree_trace.py 2> $OUT/r05_n_trace.log
  grep "^flags" $OUT/r05_n_trace.log >> $OUT/r05_n_startree.txt
  grep -B2 "^flags 0x20" $OUT/r05_n_trace.log | head -2 >> $OUT/r05_n_startree.txt
  env $knob timeout 200 python tools/startree_trace.py 2>&1 | grep "^flags" | sed 's/^/(no trace) /' >> $OUT/r05_n_startree.txt
done
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace -d $OUT/r05_n_prof -o x -- python $R/tools/startree_trace.py > /dev/null 2>&1 )
timeout 60 python tools/rocprof_summary.py $OUT/r05_n_prof/x_results.db 2>&1 | head -12 >> $OUT/r05_n_startree.txt
cat $OUT/r05_n_startree.txt | cut -c1-180
