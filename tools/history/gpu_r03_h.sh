#!/bin/bash
set -x
O=gpurun_out/r03h; mkdir -p $O; rm -f $O/ab.log
export TMPDIR=/tmp
for c in 4 3; do timeout 900 python tools/time_configs_ab.py $c trajopt_amd/_build_r02/libtrajopt_mi355x.so trajopt_amd/_build/libtrajopt_mi355x.so >> $O/ab.log 2>&1; done
cat $O/ab.log
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; grep -E "passed|failed|^FAILED" $O/pytest.log | tail -8
