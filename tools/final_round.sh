#!/bin/bash
# end-of-round evidence: gpurun -- bash tools/final_round.sh r02w
TAG=${1:-r02w}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_n1_driver_cmd.log 2> $OUT/bench_n1_driver_cmd.err
timeout 300 python bench.py > $OUT/bench_n1.log 2> $OUT/bench_n1.err
bash tools/profile_round.sh $TAG > $OUT/profile.log 2>&1
for c in 2 3 4; do timeout 300 python bench.py --config $c --steps 2 --warmup 1 > $OUT/bench_cfg$c.log 2>&1; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU -d $OUT/pmc_cfg2_mfma -o out --output-format csv -- python $R/bench.py --config 2 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_cfg2_mfma.json 2> $OUT/pmc_cfg2_mfma.log
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$OUT/pmc_cfg2_mfma/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        tot[r["Kernel_Name"][:50]][r["Counter_Name"]] += float(r["Counter_Value"])
with open("$OUT/cfg2_mfma_summary.txt", "w") as o:
    o.write("# rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU -- python bench.py --config 2 --steps 1 --warmup 0 --no-cpu-baseline\n")
    for k in tot:
        for c in sorted(tot[k]):
            o.write(f"{k:50s} {c:32s} {tot[k][c]:18.1f}\n")
print(open("$OUT/cfg2_mfma_summary.txt").read())
PY
