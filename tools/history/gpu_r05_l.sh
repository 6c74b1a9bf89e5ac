#!/bin/bash
# round 5: the two flagged cases again on the device under the harness rules that followed, then one more sweep (log only)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05n; mkdir -p $O; cd $R
sha256sum trajopt_amd/_build/libtmx_mi355x.so trajopt_amd/_build/libtrajopt_mi355x.so > $O/build_id.txt 2>/dev/null
FUZZ_ONLY=36 timeout 120 python tests/tools/fuzz_parity.py 60 91 gpu r4 lvs > $O/case_91_36_device.log 2>&1; echo "91/36 rc $?"; tail -n 1 $O/case_91_36_device.log | cut -c1-300
FUZZ_ONLY=24 timeout 120 python tests/tools/fuzz_parity.py 30 23 gpu wide > $O/case_23_24_device.log 2>&1; echo "23/24 rc $?"; tail -n 2 $O/case_23_24_device.log | cut -c1-300
t0=$SECONDS; timeout ${SW_TIMEOUT:-230} python tests/tools/fuzz_parity.py 40 101 gpu r4 lvs links > $O/fuzz_device_r4_lvs_links_40_101.log 2>&1; echo "40 101 r4 lvs links rc $? in $((SECONDS-t0)) s"
grep -v "^  note\|coredump\|execvp\|Failed to write" $O/fuzz_device_r4_lvs_links_40_101.log | tail -n 3 | cut -c1-330
