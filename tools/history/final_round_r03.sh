#!/bin/bash
# end-of-round evidence of round 3: gpurun -- bash tools/history/final_round_r03.sh r03f
TAG=${1:-r03f}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_n1_steps20.log 2> $OUT/bench_n1_steps20.err
bash tools/profile_round.sh $TAG > $OUT/profile.log 2>&1
for c in 2 3 4; do
  bash tools/profile_cfg.sh $TAG $c > $OUT/profile_cfg$c.log 2>&1
  cp $OUT/pmc_traffic_cfg$c.json $R/profiles/r03_pmc_traffic_cfg$c.json   # (on the GPU box: bench.py reads the per-config traffic from profiles/; copied again locally afterwards)
  cd $R; timeout 600 python bench.py --config $c --steps 3 --warmup 1 > $OUT/bench_cfg$c.json 2> $OUT/bench_cfg$c.err
done
[ -f trajopt_amd/_build_prof/libtrajopt_mi355x.so ] && timeout 300 python tools/prof_phases.py 1024 full trajopt_amd/_build_prof/libtrajopt_mi355x.so > $OUT/prof_phases.txt 2>&1
tail -n 3 $OUT/bench_n1_steps20.log | cut -c1-300; for c in 2 3 4; do cut -c1-1500 $OUT/bench_cfg$c.json; done; cat $OUT/prof_phases.txt
