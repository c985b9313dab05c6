#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
timeout 900 python -m pytest tests/test_gpu_headline_kernels.py -x -q -m gpu 2>&1 | tail -12
for v in "PG_WAVE_SPECIALISED=1" "PG_NO_WAVE_SPECIALISED=1"; do
  echo "== $v"
  env $v timeout 600 python tools/prof_variants.py --set cfg3 --only "group g1" --docs 1000000000 --reps 8 2>&1 | grep -v amdgpu.ids | tail -3
done
