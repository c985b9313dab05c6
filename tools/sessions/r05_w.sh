#!/bin/bash
# pg_oct_c (four load buffers) against pg_oct_l (two) and the quad kernels: parity, then the COUNT(*)-only rows
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_oct.py tests/test_gpu_headline_kernels.py -x -q -m gpu 2>&1 | tail -4
for v in "" "PG_NO_OCT_COUNT_KERNEL=1" "PG_NO_OCT_COUNT=1"; do
  echo "== ${v:-default}"
  env $v timeout 300 python tools/prof_variants.py --set cfg5 --only "count only" --reps 10 2>&1 | grep -v amdgpu.ids | tail -4
done | tee gpurun_out/r05_w_count_only.txt
