#!/bin/bash
# where the COUNT(*)-only oct kernel's time goes: SQ counters of pg_oct_l for 4 columns / 12 800 groups and for 1 column / 16 groups
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
mkdir -p gpurun_out
for q in "=cfg5 count only" "=count only group h1 (16)"; do
  echo "=== $q"
  timeout 500 python tools/pmc_sq.py cfg5 "$q" 200000000 2>&1 | grep -v amdgpu.ids | grep -A17 "^pg_oct_l"
done > gpurun_out/r05_t_sq.txt 2>&1
cat gpurun_out/r05_t_sq.txt
