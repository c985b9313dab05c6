#!/bin/bash
# the DOUBLE shapes of the wide pipeline: SQ counters (where do pg_pipe_wd_none / pg_fast_multi_wd spend their cycles)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
mkdir -p gpurun_out
for q in "=sum(d64) max(d64) group g1" "=sum(d64) avg(m) group g1"; do
  echo "=== $q"
  timeout 500 python tools/pmc_sq.py wide "$q" 100000000 2>&1 | grep -v amdgpu.ids | grep -B1 -A16 "SQ_ACTIVE_INST_ANY" | grep -v "^--"
done > gpurun_out/r05_u_sq.txt 2>&1
cat gpurun_out/r05_u_sq.txt
