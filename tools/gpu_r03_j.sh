#!/bin/bash
set -x
O=gpurun_out/r03j; mkdir -p $O; rm -f $O/ab.log
export TMPDIR=/tmp
for c in 36 37; do timeout 120 python tools/lib_smoke.py trajopt_amd/_build/libtrajopt_mi355x.so $c 2>&1 | grep -E "status|VIOLATION|fault" | head -1 | cut -c1-120; done
timeout 300 python tools/bench_libs.py 1024 trajopt_amd/_build_prev/libtrajopt_mi355x.so trajopt_amd/_build/libtrajopt_mi355x.so >> $O/ab.log 2>&1
for c in 2; do timeout 600 python tools/time_configs_ab.py $c trajopt_amd/_build_prev/libtrajopt_mi355x.so trajopt_amd/_build/libtrajopt_mi355x.so >> $O/ab.log 2>&1; done
cat $O/ab.log | cut -c1-200
