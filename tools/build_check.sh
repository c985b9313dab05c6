#!/bin/bash
# builds the library and fails loudly (use before every gpurun: a failed build silently ships the previous .so)
cd "$(dirname "$0")/../pinot_amd/csrc" && make -s -j8 2>&1 | grep -E "error|Error" && { echo "BUILD FAILED"; exit 1; }
test libpinot_gpu.so -nt pg_exec.hip -a libpinot_gpu.so -nt pg_kernels_part.hip -a libpinot_gpu.so -nt pg_plan.cpp && echo "build ok" || { echo "BUILD STALE"; exit 1; }
