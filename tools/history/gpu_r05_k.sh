#!/bin/bash
# round 5: further device sweeps on the HEAD code object with seeds no test uses (log only)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05m; mkdir -p $O; cd $R
sha256sum trajopt_amd/_build/libtrajopt_mi355x.so > $O/build_id.txt
sw() { n=$1; s=$2; shift 2; tag=$(echo "$@" | tr ' ' '_'); [ -z "$tag" ] && tag=base
  t0=$SECONDS; timeout ${SW_TIMEOUT:-170} python tests/tools/fuzz_parity.py $n $s gpu "$@" > $O/fuzz_device_${tag}_${n}_${s}.log 2>&1; echo "== $* $n $s: rc $? in $((SECONDS-t0)) s"
  grep -v "^  note\|coredump\|execvp\|Failed to write" $O/fuzz_device_${tag}_${n}_${s}.log | tail -n 3 | cut -c1-330; }
sw 60 91 r4 lvs
sw 30 97 new lvs links
sw 30 23 wide
sw 60 9
