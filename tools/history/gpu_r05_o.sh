#!/bin/bash
# round 5: what is left of the GPU budget on two more device sweeps under the final harness (log only)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05r; mkdir -p $O; cd $R
sw() { n=$1; s=$2; shift 2; tag=$(echo "$@" | tr ' ' '_'); t0=$SECONDS; timeout ${SW_TIMEOUT:-55} python tests/tools/fuzz_parity.py $n $s gpu "$@" > $O/fuzz_device_${tag}_${n}_${s}.log 2>&1; echo "== $* $n $s: rc $? in $((SECONDS-t0)) s"
  grep -v "^  note\|coredump\|execvp\|Failed to write" $O/fuzz_device_${tag}_${n}_${s}.log | tail -n 3 | cut -c1-330; }
sw 40 141 kin
sw 24 143 new lvs
