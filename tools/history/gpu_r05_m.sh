#!/bin/bash
# round 5, last call: the device tests that go through the edited harness (tests/parity_checks.py, tests/tools/fuzz_parity.py) + smoke
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05p; mkdir -p $O; cd $R
sha256sum trajopt_amd/_build/libtrajopt_mi355x.so > $O/build_id.txt
timeout 235 python -m pytest tests/test_gpu_parity.py tests/test_fuzz_parity.py -m gpu -q -x --deselect tests/test_fuzz_parity.py::test_random_problems_with_round4_features_and_pair_rows_on_device > $O/pytest_gpu_harness_subset.txt 2>&1; echo "pytest rc $?"
tail -n 3 $O/pytest_gpu_harness_subset.txt
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
