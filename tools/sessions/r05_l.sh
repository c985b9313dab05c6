#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_mv.py tests/test_mv_entry_dict.py tests/test_mv_reference_goldens.py tests/test_fuzz.py -x -q -m gpu 2>&1 | tail -8 > $OUT/r05_l_tests.txt
cat $OUT/r05_l_tests.txt
timeout 400 python tools/prof_variants.py --set mv --docs 50000000 2>&1 < /dev/null | grep -v amdgpu.ids > $OUT/r05_l_variants_mv_50m.txt
echo "== PG_MV_NO_WINDOWS=1 (the doc-by-doc walk)" >> $OUT/r05_l_variants_mv_50m.txt
PG_MV_NO_WINDOWS=1 timeout 400 python tools/prof_variants.py --set mv --docs 50000000 --only "m" 2>&1 < /dev/null | grep -v amdgpu.ids | grep "group\|summv\|avgmv\|distinct" >> $OUT/r05_l_variants_mv_50m.txt
cat $OUT/r05_l_variants_mv_50m.txt | cut -c1-150
