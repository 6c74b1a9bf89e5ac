#!/bin/bash
# end-of-round evidence of round 5 (gpurun -- bash tools/history/final_round_r05.sh r05z):
#   code-object id, the whole GPU tier, smoke; counters of configuration 1 FIRST (so that the bench line's traffic_source is this round's
#   collection), the driver's bench command, the kernel trace at the driver's command; then per configuration 2 / 3 / 4: counters, then the
#   bench line at --steps 10
TAG=${1:-r05z}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
( sha256sum trajopt_amd/_build/libtrajopt_mi355x.so; python tools/exec0_scan.py trajopt_amd/_build/libtrajopt_mi355x.so ) > $OUT/build_id.txt 2>&1
if [ -z "$SKIP_TIER" ]; then
timeout 2400 python -m pytest tests -m gpu -q -rf -rP --durations=12 > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc $?"
grep -E "passed|failed" $OUT/pytest_gpu.txt | tail -2
fi
if [ -z "$SKIP_SWEEPS" ]; then
# the device sweeps VERDICT (round 4) asked for, on this code object: 0 failures and no fault wanted
timeout 1500 python tests/tools/fuzz_parity.py 60 79 gpu r4 lvs > $OUT/fuzz_device_r4_lvs_60_79.log 2>&1; echo "60 79 r4 lvs rc $?"; tail -n 1 $OUT/fuzz_device_r4_lvs_60_79.log | cut -c1-300
timeout 1200 python tests/tools/fuzz_parity.py 40 83 gpu r4 lvs links > $OUT/fuzz_device_r4_lvs_links_40_83.log 2>&1; echo "40 83 r4 lvs links rc $?"; tail -n 1 $OUT/fuzz_device_r4_lvs_links_40_83.log | cut -c1-300
timeout 1200 python tests/tools/fuzz_parity.py 40 13 gpu new lvs > $OUT/fuzz_device_new_lvs_40_13.log 2>&1; echo "40 13 new lvs rc $?"; tail -n 1 $OUT/fuzz_device_new_lvs_40_13.log | cut -c1-300
fi
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
python tools/kernel_meta.py trajopt_amd/_build/libtrajopt_mi355x.so k_ > $OUT/kernel_meta.txt 2>&1
# counters of configuration 1 (PMC passes only), then the summary becomes this round's traffic file
SKIP_KT=1 bash tools/profile_round.sh $TAG > $OUT/profile_pmc.log 2>&1
cd $R
[ -f $OUT/pmc_traffic.json ] && cp $OUT/pmc_traffic.json profiles/r05_pmc_traffic.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_n1_steps20.log 2> $OUT/bench_n1_steps20.err
grep "^{" $OUT/bench_n1_steps20.log | cut -c1-300
# kernel trace + stats at the driver's command (summary is rewritten with both parts)
SKIP_PMC=1 bash tools/profile_round.sh $TAG > $OUT/profile_kt.log 2>&1
cd $R
python3 - <<PY
# (profile_round.sh wrote the summary in its PMC pass; append the kernel-trace part of the second call)
import csv, glob, collections
out = "$OUT"
lines = []
for f in glob.glob(out + "/kt/*kernel_stats.csv"):
    lines.append("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline")
    lines.append(open(f).read())
kt = glob.glob(out + "/kt/*kernel_trace.csv")
if kt:
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(kt[0])):
        d[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    lines.append("# per-kernel durations from the kernel trace (ms): calls, total, avg, min, max  (25 launches = 5 warm-up + 20 timed)")
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        lines.append(f"{k[:70]:70s} {len(v):5d} {sum(v):12.3f} {sum(v)/len(v):10.3f} {min(v):10.3f} {max(v):10.3f}")
        if "k_sqp_pool" in k and len(v) > 5:
            w = v[5:]
            lines.append(f"{'   ... the 20 timed launches only':70s} {len(w):5d} {sum(w):12.3f} {sum(w)/len(w):10.3f} {min(w):10.3f} {max(w):10.3f}")
prev = open(out + "/rocprofv3_summary.txt").read() if glob.glob(out + "/rocprofv3_summary.txt") else ""
open(out + "/rocprofv3_summary.txt", "w").write("\n".join(lines) + "\n" + prev)
PY
for c in ${CFGS:-2 3 4}; do
  bash tools/profile_cfg.sh $TAG $c > $OUT/profile_cfg$c.log 2>&1
  cd $R
  [ -f $OUT/pmc_traffic_cfg$c.json ] && cp $OUT/pmc_traffic_cfg$c.json profiles/r05_pmc_traffic_cfg$c.json
  timeout 900 python bench.py --config $c --steps 10 --warmup 2 > $OUT/bench_cfg$c.json 2> $OUT/bench_cfg$c.err
  grep "^{" $OUT/bench_cfg$c.json | cut -c1-260
done
head -30 $OUT/rocprofv3_summary.txt | cut -c1-200
