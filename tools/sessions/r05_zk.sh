#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
timeout 70 python -m pytest tests/test_gpu_mv.py tests/test_gpu_malformed_buffers.py -x -q -m gpu -n 4 2>&1 | tail -4
