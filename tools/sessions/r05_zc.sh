#!/bin/bash
# where pg_fast_i32range_s's time goes: the loaders' stream alone (no LDS writes, no consumers), loaders + LDS writes, the whole kernel
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
for lib in spec_stream spec_noconsume ""; do
  echo "== ${lib:-whole kernel}"
  L=""; [ -n "$lib" ] && L="PG_GPU_LIB=$R/tools/variants/libpinot_gpu_$lib.so"
  env $L PG_WAVE_SPECIALISED=1 timeout 600 python tools/prof_variants.py --set cfg3 --only "=cfg3" --docs 1000000000 --reps 10 2>&1 | grep -v amdgpu.ids | tail -1
done
