#!/bin/bash
# experiment: does the size of k_sqp_pool's private segment by itself cost time?  _build_exp = HEAD + a never-called callee with a 4 KB frame
O=gpurun_out/r03y; mkdir -p $O
A=trajopt_amd/_build/libtrajopt_mi355x.so; E=trajopt_amd/_build_exp/libtrajopt_mi355x.so
timeout 300 python tools/bench_libs.py 1024 $A $E $A $E > $O/ab_exp.log 2>&1; cat $O/ab_exp.log
