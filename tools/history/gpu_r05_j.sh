#!/bin/bash
# round 5: the several-problems-per-CU prototype (tools/ubench/btd_wave.hip), timing + one counter pass
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05k; mkdir -p $O; cd $R
timeout 120 tools/ubench/btd_wave ${ITERS:-2000} > $O/btd_wave.log 2>&1; echo "rc $?"; cat $O/btd_wave.log
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $O/pmc -o out --output-format csv -- $R/tools/ubench/btd_wave ${ITERS:-2000} > $O/pmc.log 2>&1
python3 - <<PY
import csv, glob, collections
d = collections.defaultdict(dict)
for f in glob.glob("$O/pmc/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        d[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"]); d[int(r["Dispatch_Id"])]["grid"] = r.get("Grid_Size", "")
for k in sorted(d): print(k, d[k])
PY
