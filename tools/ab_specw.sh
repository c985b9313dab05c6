#!/bin/bash
# A/B of the shared-stage frame (pg_kernels_specw.hip, PG_SPECW=1) and its variants (tools/build_variants.sh "-DSW_WAVES=8" ...) against the
# product's independent wavefronts: usage tools/ab_specw.sh "<variant> ..." [docs]   ("product" = the library in the tree, "product_w" = the tree with PG_SPECW=1,
# any other name = tools/variants/libpinot_gpu_<name>.so with PG_SPECW=1)
for v in $1; do
  echo "=== variant: $v"
  unset PG_GPU_LIB PG_SPECW
  if [ "$v" = product_w ]; then export PG_SPECW=1; elif [ "$v" != product ]; then export PG_SPECW=1 PG_GPU_LIB=$PWD/tools/variants/libpinot_gpu_$v.so; fi
  timeout 300 python tools/prof_variants.py --set dict --docs ${2:-200000000} --only "dict" 2>&1 | grep -E "^(cfg3 dict|northstar dict|cfg3 sparse|dict sel|index only)"
done
