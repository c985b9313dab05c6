#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests/test_gpu_malformed_buffers.py tests/test_gpu_admission.py tests/test_gpu_datatable.py tests/test_gpu_fake_rccl.py tests/test_gpu_multi.py tests/test_gpu_bench_contract.py -x -q -m gpu 2>&1 | tail -15 > $OUT/r05_k_tests.txt
cat $OUT/r05_k_tests.txt
timeout 1700 bash tools/sanitize.sh gpu 2>&1 | tail -8
grep -n "ERROR: AddressSanitizer\|runtime error:" $OUT/r05_sanitize_gpu.log | head -20
