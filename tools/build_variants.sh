#!/bin/bash
# Builds kernel-shape variants of libpinot_gpu.so for A/B measurements (tools/variants/libpinot_gpu_<name>.so, git-ignored but
# shipped to the GPU box); select one with PG_GPU_LIB=<path>.  usage: tools/build_variants.sh name "-DFOO=1 -DBAR=2 [PIPE_WAVES=4]" [name flags]...
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/tools/variants
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  mk=$(echo "$flags" | grep -o 'PIPE_WAVES=[0-9]*' || true)   # make variable (wavefronts per workgroup of pg_fast_i32range_p), not a -D flag
  flags=${flags/$mk/}
  d=$(mktemp -d)
  cp $R/pinot_amd/csrc/*.map $R/pinot_amd/csrc/*.hip $R/pinot_amd/csrc/*.cpp $R/pinot_amd/csrc/*.h $R/pinot_amd/csrc/*.hpp $R/pinot_amd/csrc/Makefile $d/
  mkdir -p $d/../../include $d/synth; cp $R/include/pinot_gpu.h $d/../../include/ 2>/dev/null || true
  sed -i "s#\.\./\.\./include/pinot_gpu.h#$R/include/pinot_gpu.h#" $d/pg_internal.hpp $d/Makefile
  ld=$(echo "$flags" | grep -o -- '-fsanitize=[a-z,]*' | head -1 || true)   # sanitizer variants: the runtime is linked too
  ( cd $d && make -s -j8 libpinot_gpu.so $mk CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function -munsafe-fp-atomics $flags" LDFLAGS="$ld" 2>&1 | grep -E "error|Error" || true )
  cp $d/libpinot_gpu.so $R/tools/variants/libpinot_gpu_$name.so
  grep -E "Function Name: pg_fast_i32range_a" -A 12 $d/pg_kernels.resources.log | grep -E "VGPRs:|ScratchSize|Occupancy" | sed "s/^.*remark: /  $name:/"
  rm -rf $d
done
ls -la $R/tools/variants/
