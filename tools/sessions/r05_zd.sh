#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
for lib in spec_fetchonly spec_notable; do
  echo "== ${lib}"
  env PG_GPU_LIB=$R/tools/variants/libpinot_gpu_$lib.so PG_WAVE_SPECIALISED=1 timeout 600 python tools/prof_variants.py --set cfg3 --only "=cfg3" --docs 1000000000 --reps 10 2>&1 | grep -v amdgpu.ids | tail -1
done
