#!/bin/bash
# last build of round 4: GPU tier, smoke, the driver's bench command
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04j; mkdir -p $O; cd $R
timeout 3000 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1
grep -E "passed|failed" $O/pytest_gpu.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1_steps20.log 2> $O/bench_n1_steps20.err
grep "^{" $O/bench_n1_steps20.log | cut -c1-200
