#!/bin/bash
set -x
O=gpurun_out/r03n; mkdir -p $O; rm -f $O/ab.log
export TMPDIR=/tmp
timeout 300 python tools/bench_libs.py 1024 trajopt_amd/_build_hd/libtrajopt_mi355x.so trajopt_amd/_build/libtrajopt_mi355x.so > $O/bench_libs.log 2>&1
cat $O/bench_libs.log
for c in 2 4; do timeout 600 python tools/time_configs_ab.py $c trajopt_amd/_build_hd/libtrajopt_mi355x.so trajopt_amd/_build/libtrajopt_mi355x.so >> $O/ab.log 2>&1; done
cat $O/ab.log
