#!/bin/bash
# phase profile of the banded path (config 1 + smoothing costs)
O=gpurun_out/r03q; mkdir -p $O
timeout 300 python tools/prof_phases.py 256 full trajopt_amd/_build_prof/libtrajopt_mi355x.so 1s > $O/prof_cfg1s.log 2>&1
cat $O/prof_cfg1s.log
