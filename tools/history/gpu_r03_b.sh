#!/bin/bash
# round 3, GPU call B: the whole GPU tier on the current build
set -x
O=gpurun_out/r03b; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -rP > $O/pytest_gpu.log 2>&1
tail -n 5 $O/pytest_gpu.log
grep -E "same history|config 1 x|outside 1e-5|worst|classes" $O/pytest_gpu.log | head -60
