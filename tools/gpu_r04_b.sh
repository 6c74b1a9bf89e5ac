#!/bin/bash
# A/B: r03 baseline library vs MFMA assembly of the diagonal blocks (gjm0) vs + MFMA block Gauss-Jordan (default build)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04b; mkdir -p $O; cd $R
A=trajopt_amd/_build_base/libtrajopt_mi355x.so; B=trajopt_amd/_build_gjm0/libtrajopt_mi355x.so; C=trajopt_amd/_build/libtrajopt_mi355x.so
timeout 600 python tools/bench_libs.py 1024 $A $B $C $A $B $C > $O/ab.log 2>&1; cat $O/ab.log
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -s -k "config1 or first_qp or golden" > $O/parity.log 2>&1; grep -E "passed|failed|config 1 x|outside" $O/parity.log | tail -8
