#!/bin/bash
# segmented sweeps of the dense-coupling chain (pair rows): parity of the pair-row configurations, then A/B of configurations 3 / 4 (and 1 / 2 as controls)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04g; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_sqp_flavour.py tests/test_joint_costs_kat.py -m gpu -q -x -s > $O/pytest_pairs.log 2>&1
grep -E "passed|failed|error|identical|same" $O/pytest_pairs.log | tail -14
A=trajopt_amd/_build_prev/libtrajopt_mi355x.so; C=trajopt_amd/_build/libtrajopt_mi355x.so
for c in 4 3; do timeout 900 python tools/time_configs_ab.py $c $A $C > $O/ab_cfg$c.log 2>&1; tail -3 $O/ab_cfg$c.log; done
timeout 300 python tools/bench_libs.py 1024 $A $C $A $C > $O/ab_cfg1.log 2>&1; tail -5 $O/ab_cfg1.log
timeout 600 python tools/time_configs_ab.py 2 $A $C > $O/ab_cfg2.log 2>&1; tail -3 $O/ab_cfg2.log
