#!/bin/bash
# end-of-round evidence of round 4 (gpurun -- bash tools/history/final_round_r04.sh r04x [quick]):
#   GPU tier, smoke, the driver's bench command, rocprofv3 kernel trace + ALL counter groups of configuration 1 (tools/profile_round.sh),
#   bench lines + traffic counters of configurations 2 / 3 / 4 (tools/profile_cfg.sh), phase tables on a -DTMX_PROFILE build.
TAG=${1:-r04x}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1
grep -E "passed|failed" $OUT/pytest_gpu.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $OUT/smoke.log 2>&1
tail -1 $OUT/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_n1_steps20.log 2> $OUT/bench_n1_steps20.err
grep "^{" $OUT/bench_n1_steps20.log | cut -c1-260
python tools/kernel_meta.py trajopt_amd/_build/libtrajopt_mi355x.so k_ > $OUT/kernel_meta.txt 2>&1
bash tools/profile_round.sh $TAG > $OUT/profile.log 2>&1
cd $R
for c in 2 3 4; do
  timeout 900 python bench.py --config $c --steps 3 --warmup 1 > $OUT/bench_cfg$c.json 2> $OUT/bench_cfg$c.err
  grep "^{" $OUT/bench_cfg$c.json | cut -c1-200
  [ "$2" = "quick" ] || bash tools/profile_cfg.sh $TAG $c > $OUT/profile_cfg$c.log 2>&1
  cd $R
done
if [ -f trajopt_amd/_build_prof/libtrajopt_mi355x.so ]; then
  timeout 300 python tools/prof_phases.py 1024 full trajopt_amd/_build_prof/libtrajopt_mi355x.so 1 > $OUT/prof_phases.txt 2>&1
  timeout 300 python tools/prof_phases.py 128 full trajopt_amd/_build_prof/libtrajopt_mi355x.so 3 > $OUT/prof_phases_cfg3.txt 2>&1
  timeout 300 python tools/prof_phases.py 256 full trajopt_amd/_build_prof/libtrajopt_mi355x.so 4 > $OUT/prof_phases_cfg4.txt 2>&1
  timeout 300 python tools/prof_phases.py 256 full trajopt_amd/_build_prof/libtrajopt_mi355x.so 1s > $OUT/prof_phases_cfg1_smoothing.txt 2>&1
fi
head -40 $OUT/rocprofv3_summary.txt | cut -c1-200; head -20 $OUT/prof_phases_cfg4.txt
