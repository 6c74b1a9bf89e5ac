#!/bin/bash
# time-parameterised problems on the device, then the whole GPU tier and the bench line on the same build
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04f; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_time_terms.py -m gpu -q -x -s > $O/pytest_time.log 2>&1
grep -E "passed|failed|error|^config" $O/pytest_time.log | tail -12
timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1
grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_cfg1.log 2> $O/bench_cfg1.err; python - <<'PY'
import json
for l in open('gpurun_out/r04f/bench_cfg1.log'):
    if l.startswith('{'):
        d=json.loads(l); print('bench', d['value'], d['roofline']['frac'])
PY
