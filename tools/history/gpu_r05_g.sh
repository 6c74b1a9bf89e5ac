#!/bin/bash
# A/B of a library variant against the product build on the bench workload + the three known cases + two sweeps
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${TAG:-r05g}; mkdir -p $O; cd $R
V=${1:-uni}; L=trajopt_amd/_build/v_$V/lib.so
HE=tests/hostemu/_build/libtmx_hostemu.so
timeout 600 python tools/bench_libs.py 1024 $L trajopt_amd/_build/libtrajopt_mi355x.so $L 2>&1 | grep -v "coredump\|execvp\|Failed to write" | tee $O/bench_libs_$V.log
export DIAG_ROWS="10,1,8192,1"
timeout 200 python tests/tools/diag_firstqp.py 13 11 gpu:$L $HE new lvs 2>&1 | tail -n 2 | cut -c1-170
timeout 200 python tests/tools/diag_firstqp.py 73 6 gpu:$L $HE r4 lvs 2>&1 | tail -n 2 | cut -c1-170
unset DIAG_ROWS
timeout 900 python tests/tools/fuzz_parity.py 20 73 gpu:$L r4 lvs > $O/fuzz_${V}_r4_lvs_20_73.log 2>&1; echo "rc $?"; tail -n 2 $O/fuzz_${V}_r4_lvs_20_73.log | cut -c1-300
timeout 900 python tests/tools/fuzz_parity.py 24 5 gpu:$L > $O/fuzz_${V}_24_5.log 2>&1; echo "rc $?"; tail -n 2 $O/fuzz_${V}_24_5.log | cut -c1-300
