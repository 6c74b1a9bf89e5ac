#!/bin/bash
# convex-hull links (GJK / EPA term kernels): hull + capsule geometry tests first, then the whole GPU tier and a bench line
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04e; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_hull_geometry.py tests/test_capsule_geometry.py -m gpu -q -x -s > $O/pytest_hull.log 2>&1
grep -E "passed|failed|error" $O/pytest_hull.log | tail -6
timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1
tail -5 $O/pytest_gpu.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_cfg1.log 2> $O/bench_cfg1.err; cat $O/bench_cfg1.log
