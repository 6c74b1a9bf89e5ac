#!/bin/bash
# round 5: three of the host-build cases behind the last harness rules, on the device (log only)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05q; mkdir -p $O; cd $R
one() { c=$1; n=$2; s=$3; shift 3; FUZZ_ONLY=$c timeout 60 python tests/tools/fuzz_parity.py $n $s gpu "$@" > $O/case_${s}_${c}_device.log 2>&1; echo "$s/$c rc $?"; grep -v "refused" $O/case_${s}_${c}_device.log | tail -n 2 | cut -c1-330; }
one 72 80 137 new lvs links
one 46 80 139 kin
one 52 80 131 r4 lvs
