#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
timeout 600 python -m pytest tests/test_gpu_headline_kernels.py tests/test_gpu_parity.py -x -q -m gpu -k "not specialised" 2>&1 | tail -2
for lib in "" scan3 scan2 "" scan2; do
  L=""; [ -n "$lib" ] && L="PG_GPU_LIB=$R/tools/variants/libpinot_gpu_$lib.so"
  echo "== ${lib:-4 buffers}"
  env $L timeout 300 python tools/prof_variants.py --set cfg3 --only "cfg2" --docs 100000000 --reps 20 2>&1 | grep -v amdgpu.ids | tail -1
  env $L timeout 300 python tools/prof_variants.py --set cfg3 --only "cfg2" --docs 1000000000 --reps 10 2>&1 | grep -v amdgpu.ids | tail -1
done
