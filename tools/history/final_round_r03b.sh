#!/bin/bash
# end-of-round evidence on the last commit of round 3: gpurun -- bash tools/history/final_round_r03b.sh r03x
TAG=${1:-r03x}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1
grep -E "passed|failed" $OUT/pytest_gpu.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $OUT/smoke.log 2>&1
tail -1 $OUT/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_n1_steps20.log 2> $OUT/bench_n1_steps20.err
grep "^{" $OUT/bench_n1_steps20.log | cut -c1-260
QUICK=1 bash tools/profile_round.sh $TAG > $OUT/profile.log 2>&1
cd $R
if [ -f trajopt_amd/_build_prof/libtrajopt_mi355x.so ]; then
  timeout 200 python tools/prof_phases.py 1024 full trajopt_amd/_build_prof/libtrajopt_mi355x.so 1 > $OUT/prof_phases.txt 2>&1
  timeout 200 python tools/prof_phases.py 256 full trajopt_amd/_build_prof/libtrajopt_mi355x.so 1s > $OUT/prof_phases_cfg1_smoothing.txt 2>&1
fi
head -30 $OUT/rocprofv3_summary.txt | cut -c1-200; cat $OUT/prof_phases_cfg1_smoothing.txt | head -20
