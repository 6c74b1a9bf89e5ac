#!/bin/bash
set -x
O=gpurun_out/r03j; mkdir -p $O; rm -f $O/ab.log
export TMPDIR=/tmp
for c in 2 4 3; do timeout 600 python tools/time_configs_ab.py $c trajopt_amd/_build_prev/libtrajopt_mi355x.so trajopt_amd/_build/libtrajopt_mi355x.so >> $O/ab.log 2>&1; done
timeout 300 python tools/bench_libs.py 1024 trajopt_amd/_build_prev/libtrajopt_mi355x.so trajopt_amd/_build/libtrajopt_mi355x.so >> $O/ab.log 2>&1
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
cat $O/ab.log | cut -c1-200
