#!/bin/bash
# round 3, end: the whole GPU tier, smoke() and the bench at the driver's command
O=gpurun_out/r03u; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1
tail -5 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.log 2>&1
tail -2 $O/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1_steps20.log 2> $O/bench_n1_steps20.err
tail -n 1 $O/bench_n1_steps20.log | cut -c1-900
