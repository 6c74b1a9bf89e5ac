#!/bin/bash
# round 5: the whole GPU tier (no -x: every test runs), smoke and the driver's bench command on the product build
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${TAG:-r05f}; mkdir -p $O; cd $R
( sha256sum trajopt_amd/_build/libtrajopt_mi355x.so; git rev-parse HEAD 2>/dev/null ) > $O/build_id.txt
timeout 2400 python -m pytest tests -m gpu -q -rf -rP --durations=15 > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?"; tail -n 30 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -n 2 $O/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1_steps20.log 2>&1; echo "bench rc $?"; tail -n 1 $O/bench_n1_steps20.log
