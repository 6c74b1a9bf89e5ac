#!/bin/bash
set -x
O=gpurun_out/r03i; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "acc_jerk or 26 or 27 or 28 or joint_vel_kat or joint_pos_kat" > $O/pytest_acc.log 2>&1
tail -5 $O/pytest_acc.log
timeout 300 python tools/bench_libs.py 1024 trajopt_amd/_build_prev/libtrajopt_mi355x.so trajopt_amd/_build/libtrajopt_mi355x.so trajopt_amd/_build_prev/libtrajopt_mi355x.so trajopt_amd/_build/libtrajopt_mi355x.so > $O/bench_libs.log 2>&1
cat $O/bench_libs.log
for c in 4 3; do timeout 600 python tools/time_configs_ab.py $c trajopt_amd/_build_prev/libtrajopt_mi355x.so trajopt_amd/_build/libtrajopt_mi355x.so >> $O/ab.log 2>&1; done
cat $O/ab.log
