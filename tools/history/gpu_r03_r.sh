#!/bin/bash
# round 3: built-in kinematic function terms (AvoidSingularity, DynamicCartPose) on the device + banded path at the baseline batch
O=gpurun_out/r03r; mkdir -p $O
timeout 600 python -m pytest tests/test_kinematic_terms.py tests/test_gpu_parity.py tests/test_cpp_host_api.py -m gpu -q -x -k "kinematic or 36 or 37 or 38 or 39 or 40 or 41 or cpp" > $O/pytest.log 2>&1
tail -5 $O/pytest.log
timeout 300 python tools/time_acc_jerk.py 1024 > $O/time_acc_jerk.log 2>&1
cat $O/time_acc_jerk.log
