#!/bin/bash
# HBM traffic counters of the bench kernel only (two separate PMC passes): gpurun -- bash tools/pmc_traffic.sh <tag> [bench args]
TAG=${1:-r02x}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for grp in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $grp -d $OUT/pmc_$grp -o out --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $OUT/pmc_$grp.json 2> $OUT/pmc_$grp.log
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$OUT/pmc_*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        tot[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Kernel_Name"]][r["Counter_Name"]] += 1
for k in tot:
    for c in sorted(tot[k]):
        print(f"{k[:50]:50s} {c:12s} {cnt[k][c]:4d} launches  {tot[k][c]/cnt[k][c]/1048576.0:10.3f} GiB per launch (counter in KiB)")
PY
