#!/bin/bash
# round 3: kinematic built-ins on the device (tests + randomised sweep) and a timing sanity of the four BASELINE configurations
O=gpurun_out/r03s; mkdir -p $O
timeout 900 python -m pytest tests/test_kinematic_terms.py tests/test_fuzz_parity.py tests/test_gpu_parity.py -m gpu -q -x -k "kinematic or 38 or 39 or 40 or 41 or 42 or 43" > $O/pytest.log 2>&1
tail -4 $O/pytest.log
timeout 600 python tools/time_configs.py 1 2 3 4 > $O/time_configs.log 2>&1
cat $O/time_configs.log
