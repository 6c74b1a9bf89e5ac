#!/bin/bash
# round 3, GPU call C: epoch-resident burst vs the check-per-burst build: results, throughput, phase table
set -x
O=gpurun_out/r03c; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/bench_libs.py 1024 trajopt_amd/_build/libtrajopt_mi355x.so trajopt_amd/_build_ep/libtrajopt_mi355x.so > $O/bench_libs.log 2>&1
timeout 300 python tools/prof_phases.py 1024 full trajopt_amd/_build_epp/libtrajopt_mi355x.so > $O/prof_ep.log 2>&1
cat $O/bench_libs.log $O/prof_ep.log
