#!/bin/bash
# COUNT(*)-only group-bys in the oct layout: parity, then A/B against the quad kernels
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_oct.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -4
for v in "" "PG_NO_OCT_COUNT=1"; do
  echo "== ${v:-default}"
  env $v timeout 300 python tools/prof_variants.py --set cfg5 --only "count only" --reps 10 2>&1 | grep -v amdgpu.ids | tail -8
done
