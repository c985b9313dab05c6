#!/bin/bash
# Prints VGPR / spill / scratch usage per kernel of pg_kernels.hip (dev tool). Extra args are passed to hipcc.
cd "$(dirname "$0")/../pinot_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -save-temps=obj -Rpass-analysis=kernel-resource-usage "$@" -c pg_kernels.hip -o /tmp/k.o 2>&1 |
  grep -E "Function Name|VGPRs:|VGPRs Spill|SGPRs Spill|ScratchSize|TotalSGPRs" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | paste - - - - - -
