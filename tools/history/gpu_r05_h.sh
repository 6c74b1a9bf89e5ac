#!/bin/bash
# round 5, last device session: the three device sweeps of VERDICT (round 4) item 1 again, on the HEAD code object (log only)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05h2; mkdir -p $O; cd $R
sha256sum trajopt_amd/_build/libtrajopt_mi355x.so > $O/build_id.txt
sw() { n=$1; s=$2; shift 2; tag=$(echo "$@" | tr ' ' '_')
  t0=$SECONDS; timeout ${SW_TIMEOUT:-330} python tests/tools/fuzz_parity.py $n $s gpu "$@" > $O/fuzz_device_${tag}_${n}_${s}.log 2>&1; echo "== $* $n $s: rc $? in $((SECONDS-t0)) s"
  grep -v "^  note\|coredump\|execvp\|Failed to write" $O/fuzz_device_${tag}_${n}_${s}.log | tail -n 4 | cut -c1-300; }
sw 60 79 r4 lvs
sw 40 83 r4 lvs links
sw 40 13 new lvs
