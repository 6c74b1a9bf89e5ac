#!/bin/bash
# round 3: row-only function terms on the structured solvers (no size limit): kinematic + function-term tests and sweeps on the device
O=gpurun_out/r03t; mkdir -p $O
timeout 900 python -m pytest tests/test_kinematic_terms.py tests/test_func_terms.py tests/test_fuzz_parity.py tests/test_gpu_parity.py tests/test_cpp_host_api.py -m gpu -q -x -k "kinematic or func or user_defined or small_problems or round3 or 38 or 39 or 40 or 41 or 42 or 43 or 44 or cpp" > $O/pytest.log 2>&1
tail -4 $O/pytest.log
