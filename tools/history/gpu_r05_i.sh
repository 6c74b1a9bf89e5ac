#!/bin/bash
# round 5: configurations 2 and 4 on the HEAD code object - counters first, then the bench line at --steps 10 (as final_round_r05.sh does)
TAG=r05z2; R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
sha256sum trajopt_amd/_build/libtrajopt_mi355x.so > $OUT/build_id.txt
for c in ${CFGS:-4 2}; do
  bash tools/profile_cfg.sh $TAG $c > $OUT/profile_cfg$c.log 2>&1
  cd $R
  [ -f $OUT/pmc_traffic_cfg$c.json ] && cp $OUT/pmc_traffic_cfg$c.json profiles/r05_pmc_traffic_cfg$c.json
  timeout 400 python bench.py --config $c --steps 10 --warmup 2 > $OUT/bench_cfg$c.json 2> $OUT/bench_cfg$c.err
  grep "^{" $OUT/bench_cfg$c.json | cut -c1-260
done
