#!/bin/bash
# a14: difference rows of order 2 / 3 on the banded structured path - GPU tier of the joint-cost tests, A/B of the configurations against the previous build
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04d; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_joint_costs_kat.py -m gpu -q -x -s > $O/pytest_joint.log 2>&1
grep -E "passed|failed|error|config 1 \+" $O/pytest_joint.log | tail -6
A=trajopt_amd/_build_gjm0/libtrajopt_mi355x.so; C=trajopt_amd/_build/libtrajopt_mi355x.so
timeout 300 python tools/bench_libs.py 1024 $A $C $A $C > $O/ab_cfg1.log 2>&1; cat $O/ab_cfg1.log
for c in 2 3 4; do timeout 600 python tools/time_configs_ab.py $c $A $C > $O/ab_cfg$c.log 2>&1; tail -4 $O/ab_cfg$c.log; done
