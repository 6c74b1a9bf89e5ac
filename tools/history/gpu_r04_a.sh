#!/bin/bash
# round 4, first call: GPU tier on the hygiene commit + smoke + the driver's bench command
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04a; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 > $O/pytest_gpu.log 2>&1
grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3
grep -E "config [23] x|smoothing|acc \+ jerk" $O/pytest_gpu.log | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1_steps20.log 2> $O/bench_n1_steps20.err
grep "^{" $O/bench_n1_steps20.log | cut -c1-400
