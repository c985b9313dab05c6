#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
for knob in "PG_X=1" "PG_NO_FUSED_FINISH=1"; do
  echo "# $knob"
  env $knob PG_TRACE_HOST=1 timeout 200 python tools/startree_trace.py 2> $OUT/r05_o_trace.log
  grep -B3 "^flags 0x20" $OUT/r05_o_trace.log | head -4
done
