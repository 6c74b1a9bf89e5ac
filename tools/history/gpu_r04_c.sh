#!/bin/bash
# GPU tier + class statistics + bench on the build with the D x D diagonal blocks of the KKT factorisation assembled on the matrix cores
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04c; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q -x -s > $O/pytest_gpu.log 2>&1
grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3
grep -E "config [1234] x|outside 1e-5|same history|acc \+ jerk" $O/pytest_gpu.log | head -30
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1_steps20.log 2> $O/bench_n1_steps20.err
grep "^{" $O/bench_n1_steps20.log | cut -c1-200
