#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_headline_kernels.py -x -q -m gpu -k "specialised" 2>&1 | tail -3
PG_WAVE_SPECIALISED=1 timeout 900 python tools/pmc_sq.py cfg3 "=cfg3" 1000000000 2>&1 | grep -v amdgpu.ids | grep -A17 "^pg_fast_i32range_s" | tee gpurun_out/r05_za_sq_spec.txt
