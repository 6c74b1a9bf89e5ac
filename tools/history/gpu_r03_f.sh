#!/bin/bash
set -x
O=gpurun_out/r03f; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/prof_phases.py 1024 full trajopt_amd/_build_prof/libtrajopt_mi355x.so 1 > $O/prof_cfg1.log 2>&1
timeout 300 python tools/prof_phases.py 128 full trajopt_amd/_build_prof/libtrajopt_mi355x.so 3 > $O/prof_cfg3.log 2>&1
timeout 300 python tools/prof_phases.py 256 full trajopt_amd/_build_prof/libtrajopt_mi355x.so 4 > $O/prof_cfg4.log 2>&1
cat $O/prof_cfg1.log $O/prof_cfg3.log $O/prof_cfg4.log
