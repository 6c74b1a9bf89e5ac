#!/bin/bash
# randomised parity sweep with the round-4 term families on the device (convex-hull / capsule links under all evaluators, time-parameterised problems)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04k; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_fuzz_parity.py -m gpu -q -x -k round4 > $O/pytest_fuzz_r4.log 2>&1; tail -2 $O/pytest_fuzz_r4.log
timeout 1500 python tests/tools/fuzz_parity.py 60 79 gpu r4 lvs > $O/fuzz_device_r4_60.log 2>&1; tail -3 $O/fuzz_device_r4_60.log
timeout 900 python tests/tools/fuzz_parity.py 40 83 gpu r4 lvs links > $O/fuzz_device_r4_links_40.log 2>&1; tail -3 $O/fuzz_device_r4_links_40.log
