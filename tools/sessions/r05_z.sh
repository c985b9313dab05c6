#!/bin/bash
# pg_fast_i32range_s (wave-specialised) against pg_fast_i32range_p: parity, then config 3 / north-star over 10^9 docs
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_headline_kernels.py -x -q -m gpu -k "specialised" 2>&1 | tail -15
for v in "PG_WAVE_SPECIALISED=1" ""; do
  echo "== ${v:-default}"
  env $v timeout 600 python tools/prof_variants.py --set cfg3 --only "=cfg3" --docs 1000000000 --reps 10 2>&1 | grep -v amdgpu.ids | tail -1
  env $v timeout 600 python tools/prof_variants.py --set cfg3 --only "=northstar" --docs 1000000000 --reps 10 2>&1 | grep -v amdgpu.ids | tail -1
done | tee gpurun_out/r05_z_wave_specialised.txt
