#!/bin/bash
set -x
O=gpurun_out/r03j; mkdir -p $O; rm -f $O/ab.log
export TMPDIR=/tmp
for c in 4 3; do timeout 600 python tools/time_configs_ab.py $c trajopt_amd/_build_prev/libtrajopt_mi355x.so trajopt_amd/_build/libtrajopt_mi355x.so >> $O/ab.log 2>&1; done
timeout 900 python -m pytest tests -m gpu -q -x -k "16 or 17 or 18 or 19 or 21 or 22 or 23 or 24 or 25 or flavour or fuzz" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
cat $O/ab.log
