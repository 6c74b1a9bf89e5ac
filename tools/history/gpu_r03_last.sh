#!/bin/bash
# last library build of round 3: the bench lines of configurations 4 and 3 (bench.py --config N)
O=gpurun_out/r03last; mkdir -p $O
for c in 4 3; do timeout 200 python bench.py --config $c --steps 3 --warmup 1 > $O/bench_cfg$c.json 2> $O/bench_cfg$c.err; grep "^{" $O/bench_cfg$c.json | cut -c1-330; done
