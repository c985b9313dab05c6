#!/bin/bash
# DOUBLE shapes of the wide pipeline after the doc-by-doc aggregation: parity, then the wide variants table
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline_kernels.py tests/test_gpu_oct.py -x -q -m gpu 2>&1 | tail -4
timeout 600 python tools/prof_variants.py --set wide --docs 100000000 --reps 10 2>&1 | grep -v amdgpu.ids | tail -12 | tee gpurun_out/r05_v_variants_wide_100m.txt
