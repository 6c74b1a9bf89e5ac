#!/bin/bash
# HBM-traffic counters of one `bench.py --config N` launch (run on the GPU box: bash tools/profile_cfg.sh <tag> <config>):
# separate rocprofv3 --pmc passes for FETCH_SIZE and WRITE_SIZE (kernel trace only, as the MI355X guide prescribes), summed over
# the dominant kernel's dispatches of a one-step run -> gpurun_out/<tag>/pmc_traffic_cfg<N>.json
TAG=${1:-r03}
CFG=${2:-2}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for grp in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 400 rocprofv3 --kernel-trace --pmc $grp -d $OUT/pmc_cfg${CFG}_$tag -o out --output-format csv -- python $R/bench.py --config $CFG --steps 1 --warmup 0 --depth 1 --no-cpu-baseline > $OUT/pmc_cfg${CFG}_$tag.json 2> $OUT/pmc_cfg${CFG}_$tag.log
done
python3 - <<PY
import csv, glob, json, collections
out, cfg = "$OUT", int("$CFG")
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out + f"/pmc_cfg{cfg}_*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        tot[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Kernel_Name"]][r["Counter_Name"]] += 1
lines = [f"# rocprofv3 --kernel-trace --pmc <group> -- python bench.py --config {cfg} --steps 1 --warmup 0 --depth 1 --no-cpu-baseline",
         "# kernel, counter, dispatches, sum, avg per dispatch"]
for k in tot:
    for c in sorted(tot[k]):
        lines.append(f"{k[:60]:60s} {c:32s} {cnt[k][c]:5d} {tot[k][c]:18.1f} {tot[k][c]/cnt[k][c]:18.1f}")
open(out + f"/cfg{cfg}_pmc_summary.txt", "w").write("\n".join(lines) + "\n")
main = max(tot, key=lambda k: tot[k].get("SQ_WAVE_CYCLES", 0.0) + tot[k].get("FETCH_SIZE", 0.0))
b = json.loads(next(l for l in open(out + f"/pmc_cfg{cfg}_FETCH_SIZE.json") if l.startswith("{")))
fetch_kib = tot[main].get("FETCH_SIZE", 0.0) / max(1, cnt[main].get("FETCH_SIZE", 1))
write_kib = tot[main].get("WRITE_SIZE", 0.0) / max(1, cnt[main].get("WRITE_SIZE", 1))
json.dump({"batch_per_gpu": b["config"]["batch_per_gpu"], "config": cfg, "kernel": main.split("(")[0], "fetch_size_kib_per_launch": fetch_kib,
           "write_size_kib_per_launch": write_kib, "hbm_bytes_per_launch": (2.0 * fetch_kib + write_kib) * 1024.0,
           "note": "FETCH_SIZE doubled per the MI355X guide's gfx950 correction for wide streaming reads (upper bound for narrow reads); "
                   "WRITE_SIZE uncalibrated; Infinity-Cache hits are counted"}, open(out + f"/pmc_traffic_cfg{cfg}.json", "w"), indent=1)
print("\n".join(lines)); print(open(out + f"/pmc_traffic_cfg{cfg}.json").read())
PY
