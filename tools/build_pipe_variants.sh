#!/bin/bash
# Variants of pg_kernels_pipe.hip only (seconds each), linked against the in-tree objects: tools/variants/libpinot_gpu_<name>.so.
# usage: tools/build_pipe_variants.sh name "-DPG_WAVES_PER_BLOCK=4 -DPG_PIPE_BLOCKED" [name flags]...
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/pinot_amd/csrc
mkdir -p $R/tools/variants
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  o=$(mktemp --suffix=.o)
  case "$flags" in *PG_WAVES_PER_BLOCK*) ;; *) flags="$flags -DPG_WAVES_PER_BLOCK=8";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -munsafe-fp-atomics $flags -Rpass-analysis=kernel-resource-usage -c $C/pg_kernels_pipe.hip -o $o 2> $o.log || { cat $o.log; exit 1; }
  grep -A12 "Function Name: pg_fast_i32range_p" $o.log | grep -E "VGPRs:|ScratchSize" | sed "s/^.*remark: /  $name:/"
  objs=$(cd $C && ls *.o | grep -v pg_kernels_pipe.o | sed "s#^#$C/#")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/variants/libpinot_gpu_$name.so $objs $o -ldl
  rm -f $o $o.log
done
