#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_group_trim.py tests/test_host_formats.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -12 > $OUT/r05_i_tests.txt
cat $OUT/r05_i_tests.txt
timeout 300 python tools/trim_latency.py 100000000 2>&1 | grep -v amdgpu.ids | tee $OUT/r05_i_trim_latency.txt
PG_TRACE_HOST=1 timeout 200 python tools/prof_variants.py --set general --docs 200000000 --only "cfg3 filter, 160k" --reps 2 2>&1 | grep -v amdgpu.ids | grep "limit by prefix\|160k" | head -8 | tee $OUT/r05_i_prefix_trace.txt
