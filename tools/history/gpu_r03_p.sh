#!/bin/bash
# refresh of the configuration 2 / 3 / 4 evidence after the last generic-path commits (tools/history/final_round_r03.sh layout)
TAG=r03h
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_n1_steps20.log 2> $OUT/bench_n1_steps20.err
for c in 2 3 4; do
  bash tools/profile_cfg.sh $TAG $c > $OUT/profile_cfg$c.log 2>&1
  cp $OUT/pmc_traffic_cfg$c.json $R/profiles/r03_pmc_traffic_cfg$c.json
  cd $R; timeout 600 python bench.py --config $c --steps 3 --warmup 1 > $OUT/bench_cfg$c.json 2> $OUT/bench_cfg$c.err
done
cd trajopt_amd/csrc >/dev/null; cd $R
for c in 3 4; do timeout 300 python tools/prof_phases.py $([ $c = 3 ] && echo 128 || echo 256) full trajopt_amd/_build_prof/libtrajopt_mi355x.so $c > $OUT/prof_cfg$c.log 2>&1; done
timeout 600 python tests/tools/c4_parity_stat.py 32 > $OUT/c4_parity_32.log 2>&1
grep "^{" $OUT/bench_n1_steps20.log | cut -c1-300; for c in 2 3 4; do grep "^{" $OUT/bench_cfg$c.json | cut -c1-400; done; tail -3 $OUT/c4_parity_32.log
