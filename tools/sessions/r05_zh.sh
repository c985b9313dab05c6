#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
timeout 900 python -m pytest tests/test_gpu_headline_kernels.py -x -q -m gpu 2>&1 | tail -8
for v in "" "PG_NO_WAVE_SPECIALISED=1"; do
  echo "== ${v:-default}"
  env $v timeout 600 python tools/prof_variants.py --set upsert --only "=cfg3" --docs 1000000000 --reps 8 2>&1 | grep -v amdgpu.ids | tail -1
  env $v timeout 600 python tools/prof_variants.py --set upsert --only "=northstar" --docs 1000000000 --reps 8 2>&1 | grep -v amdgpu.ids | tail -1
done
