#!/bin/bash
set -x
O=gpurun_out/r03k; mkdir -p $O; rm -f $O/ab.log
export TMPDIR=/tmp
timeout 600 python tools/time_configs_ab.py 2 trajopt_amd/_build_prev/libtrajopt_mi355x.so trajopt_amd/_build_head/libtrajopt_mi355x.so trajopt_amd/_build/libtrajopt_mi355x.so trajopt_amd/_build_prev/libtrajopt_mi355x.so >> $O/ab.log 2>&1
cat $O/ab.log
