#!/bin/bash
# A/B of configuration 1: previous build vs the build with the polish of the fast path out of line (bit-identity is reported by the tool)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04i; mkdir -p $O; cd $R
A=trajopt_amd/_build_prev/libtrajopt_mi355x.so; C=trajopt_amd/_build/libtrajopt_mi355x.so
timeout 400 python tools/bench_libs.py 1024 $A $C $A $C > $O/ab_cfg1.log 2>&1; cat $O/ab_cfg1.log
for c in 2 4; do timeout 600 python tools/time_configs_ab.py $c $A $C > $O/ab_cfg$c.log 2>&1; tail -2 $O/ab_cfg$c.log; done
