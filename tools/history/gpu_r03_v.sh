#!/bin/bash
# same-box bisection of the config-1 rate over this session's commits (sources of each commit against the current tmx.h)
O=gpurun_out/r03v; mkdir -p $O
L=""; for c in prev ed0a149 2ce379a c5ca9ef f62d297 29926e3 9c900b5; do L="$L trajopt_amd/_build_$c/libtrajopt_mi355x.so"; done
timeout 600 python tools/bench_libs.py 1024 $L trajopt_amd/_build/libtrajopt_mi355x.so trajopt_amd/_build_prev/libtrajopt_mi355x.so > $O/ab_cfg1.log 2>&1
cat $O/ab_cfg1.log
