#!/bin/bash
# whole GPU tier on the build with segmented dense-coupling sweeps
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04h; mkdir -p $O; cd $R
timeout 3000 python -m pytest tests -m gpu -q -x -s > $O/pytest_gpu.log 2>&1
grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3
grep -E "oracle vs FMA oracle: ADMM|config 3 x 32|config 1 x 64|config 2 x 16" $O/pytest_gpu.log | tail
