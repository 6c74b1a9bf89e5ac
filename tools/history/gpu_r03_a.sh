#!/bin/bash
# round 3, GPU call A: baseline bench, 2-workgroups-per-CU build experiment, config 1 / config 4 history statistics
set -x
O=gpurun_out/r03a; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tests/tools/c1_parity_stat.py 64 > $O/c1_parity_64.log 2>&1
timeout 600 python tests/tools/c4_parity_stat.py 32 > $O/c4_parity_32.log 2>&1
timeout 300 python tools/bench_libs.py 1024 trajopt_amd/_build/libtrajopt_mi355x.so trajopt_amd/_build_w2/libtrajopt_mi355x.so > $O/bench_libs.log 2>&1
timeout 300 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
tail -n 30 $O/*.log $O/bench.json
