#!/bin/bash
# round 3, GPU call D: GPU tier + A/B of the epoch-resident burst (A B A B on one box) + bench line
set -x
O=gpurun_out/r03d; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -rP > $O/pytest_gpu.log 2>&1
tail -n 3 $O/pytest_gpu.log
timeout 600 python tools/bench_libs.py 1024 trajopt_amd/_build_base/libtrajopt_mi355x.so trajopt_amd/_build/libtrajopt_mi355x.so trajopt_amd/_build_base/libtrajopt_mi355x.so trajopt_amd/_build/libtrajopt_mi355x.so > $O/bench_libs.log 2>&1
cat $O/bench_libs.log
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
cut -c1-400 $O/bench.json
