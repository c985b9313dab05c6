#!/bin/bash
# SQ counters of the headline query through pg_fast_i32range_s (PG_WAVE_SPECIALISED=1) and pg_fast_i32range_p, 10^9 docs
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
mkdir -p gpurun_out
{
echo "=== PG_WAVE_SPECIALISED=1"
PG_WAVE_SPECIALISED=1 timeout 900 python tools/pmc_sq.py cfg3 "=cfg3" 1000000000 2>&1 | grep -v amdgpu.ids | grep -A16 "^pg_fast_i32range_s"
echo "=== default"
timeout 900 python tools/pmc_sq.py cfg3 "=cfg3" 1000000000 2>&1 | grep -v amdgpu.ids | grep -A16 "^pg_fast_i32range_p"
} | tee gpurun_out/r05_zb_sq.txt
