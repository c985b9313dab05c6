#!/bin/bash
# A/B of pg_kernels_specd.hip variants (tools/build_variants.sh): usage tools/ab_specd.sh "<variant>[:wgs_per_cu] ..." [docs]   ("product" = the library in the tree)
for spec in $1; do
  v=${spec%%:*}; w=${spec##*:}; [ "$w" = "$spec" ] && w=1
  echo "=== variant: $v, PG_SPECD_WGS_PER_CU=$w"
  if [ "$v" != product ]; then export PG_GPU_LIB=$PWD/tools/variants/libpinot_gpu_$v.so; else unset PG_GPU_LIB; fi
  PG_SPECD_WGS_PER_CU=$w timeout 300 python tools/prof_variants.py --set dict --docs ${2:-200000000} --only "dict" 2>&1 | grep -E "^(cfg3 dict|northstar dict|no filter sum\(m_d\)|dict sel)"
done
