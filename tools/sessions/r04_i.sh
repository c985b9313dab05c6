#!/bin/bash
# round 4, session i: dword registers in pg_oct_l (ds_max_u32) A/B; variants rows + kernel stats for multi-value kernels, raw STRING group-by, final DISTINCT values
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O; TAG=${1:-r04_i}
timeout 300 python -m pytest tests/test_gpu_oct.py -q -p no:cacheprovider -x --timeout 120 > $O/${TAG}_oct.log 2>&1; echo "oct tests rc=$?"; tail -4 $O/${TAG}_oct.log | cut -c1-300
timeout 400 python tools/prof_variants.py --set cfg5 --docs 200000000 --reps 5 2>&1 | grep -v "^/opt" | tee $O/${TAG}_variants_cfg5_200m.txt
echo "== PG_OCT_BYTE_REGS=1"; PG_OCT_BYTE_REGS=1 timeout 200 python tools/prof_variants.py --set cfg5 --only "hll(u)" --docs 200000000 --reps 5 2>&1 | grep -v "^/opt" | tee $O/${TAG}_variants_cfg5_byte_regs.txt
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_mv; timeout 500 rocprofv3 --kernel-trace -d /tmp/prof_mv -o x -- python $R/tools/prof_variants.py --set mv --docs 50000000 --reps 5 2>&1 | grep -v "^/opt" | grep -v "^W2\|^E2\|rocprof" | tee $O/${TAG}_variants_mv_50m.txt
db=$(find /tmp/prof_mv -name "*_results.db" | head -1); python $R/tools/rocprof_summary.py $db > $O/${TAG}_kernels_mv.txt 2>&1; head -14 $O/${TAG}_kernels_mv.txt | cut -c1-150
cd $R
timeout 300 python tools/prof_variants.py --set strings --docs 20000000 --reps 5 2>&1 | grep -v "^/opt" | tee $O/${TAG}_variants_strings_20m.txt
