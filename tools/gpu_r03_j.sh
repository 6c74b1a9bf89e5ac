#!/bin/bash
set -x
O=gpurun_out/r03j; mkdir -p $O; rm -f $O/ab.log
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "31 or 32 or 33 or 34 or 35 or capsule or box or mesh" > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2
timeout 300 python tools/bench_libs.py 1024 trajopt_amd/_build_prev/libtrajopt_mi355x.so trajopt_amd/_build/libtrajopt_mi355x.so >> $O/ab.log 2>&1
for c in 3; do timeout 600 python tools/time_configs_ab.py $c trajopt_amd/_build_prev/libtrajopt_mi355x.so trajopt_amd/_build/libtrajopt_mi355x.so >> $O/ab.log 2>&1; done
cat $O/ab.log | cut -c1-200
