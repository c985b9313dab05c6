#!/bin/bash
# pruned passes with three sized load buffers: parity, then config 5 at 2 x 10^8 and 10^9 docs
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_oct.py tests/test_gpu_partition_pipeline.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python tools/prof_variants.py --set cfg5 --only "=cfg5" --docs 200000000 --reps 10 2>&1 | grep -v amdgpu.ids | tail -2
timeout 600 python tools/prof_variants.py --set cfg5 --only "=cfg5" --docs 1000000000 --reps 10 2>&1 | grep -v amdgpu.ids | tail -2
