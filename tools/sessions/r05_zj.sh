#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_fuzz.py tests/test_gpu_headline_kernels.py tests/test_null_and_valid_docs.py -x -q -m gpu -n 4 2>&1 | tail -3
for v in "" "PG_NO_PIPE_NO_GROUP=1"; do
  echo "== ${v:-default}"
  env $v timeout 300 python tools/prof_variants.py --set cfg3 --only "no " --docs 1000000000 --reps 8 2>&1 | grep -v amdgpu.ids | grep "no filter sum(m)  \|no group" 
done
