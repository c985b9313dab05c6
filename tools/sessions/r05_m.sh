#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_final_distinct.py tests/test_gpu_startree.py tests/test_gpu_oct.py tests/test_gpu_parity.py tests/test_gpu_soak.py tests/test_combine_threads.py -x -q -m gpu 2>&1 | tail -6 > $OUT/r05_m_tests.txt
cat $OUT/r05_m_tests.txt
for knob in "PG_X=1" "PG_NO_FUSED_FINISH=1"; do
  echo "# $knob" >> $OUT/r05_m_startree.txt
  env $knob PG_TRACE_HOST=1 timeout 200 python tools/startree_trace.py 2>&1 | grep -v amdgpu.ids | tail -12 >> $OUT/r05_m_startree.txt
done
cat $OUT/r05_m_startree.txt | cut -c1-200
