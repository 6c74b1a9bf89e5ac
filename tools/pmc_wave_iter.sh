#!/bin/bash
# PMC counters of the wave-pair ADMM loop in isolation (tools/wave_iter_cost.py: one burst of 2000 iterations per problem)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-r06_pmc_iter}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --kernel-trace --pmc $grp -d $OUT/pmc_$tag -o out --output-format csv -- python $R/tools/wave_iter_cost.py 1024 500 > $OUT/pmc_$tag.log 2>&1
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$OUT/pmc_*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        tot[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
for k in tot:
    if "wave" in k:
        for c in sorted(tot[k]):
            print(f"{k[:40]:40s} {c:28s} {tot[k][c]:18.0f}")
PY
