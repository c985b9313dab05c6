R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "WRITE_SIZE" "FETCH_SIZE"; do
  n=$(echo $set | cut -d" " -f1)
  timeout 45 rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmcs_$n -o x -- python $R/tools/prof_variants.py --set cfg5 --docs 200000000 --reps 2 --only "${1:-cfg5}" > /dev/null 2>&1 < /dev/null
  echo "== $set"
  timeout 20 python $R/tools/rocprof_summary.py $R/gpurun_out/pmcs_$n/x_results.db < /dev/null | grep -A30 "^counters" | grep "pg_radix\|pg_fast_none" | head -12
done
