#!/bin/bash
set -x
O=gpurun_out/r03g; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "3 or config3 or config2 or 2-" > $O/pytest_cfg3.log 2>&1
tail -n 3 $O/pytest_cfg3.log
for c in 3 2 4; do timeout 600 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_cfg$c.json 2> $O/bench_cfg$c.err; cut -c1-330 $O/bench_cfg$c.json | head -1; done
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json | head -1
