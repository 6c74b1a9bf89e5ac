#!/bin/bash
# same-box A/B after the band / no-band kernel split: pre-session library (f9abc0f sources) vs HEAD on configs 1 - 4; banded tests
O=gpurun_out/r03w; mkdir -p $O
A=trajopt_amd/_build_prev/libtrajopt_mi355x.so; Bn=trajopt_amd/_build/libtrajopt_mi355x.so
timeout 300 python tools/bench_libs.py 1024 $A $Bn $A $Bn > $O/ab_cfg1.log 2>&1; cat $O/ab_cfg1.log
for c in 2 3 4; do timeout 300 python tools/time_configs_ab.py $c $A $Bn > $O/ab_cfg$c.log 2>&1; tail -3 $O/ab_cfg$c.log; done
timeout 600 python -m pytest tests -m gpu -q -x -k "36 or 37 or smoothing" > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2
timeout 200 python tools/time_acc_jerk.py 256 > $O/time_acc_jerk.log 2>&1; cat $O/time_acc_jerk.log
