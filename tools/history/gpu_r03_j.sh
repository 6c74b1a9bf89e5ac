#!/bin/bash
set -x
O=gpurun_out/r03j; mkdir -p $O; rm -f $O/ab.log
export TMPDIR=/tmp

timeout 300 python - <<'PY' 2>&1 | tail -6
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from trajopt_amd import configs, abi, runtime
from trajopt_amd.problem import JointAccTermInfo, JointJerkTermInfo
pci, s, g = configs.config1()
pci.cost_infos.append(JointAccTermInfo(coeffs=[1.0] * 7, targets=[0.0] * 7, first_step=0, last_step=29, name="acc"))
pci.cost_infos.append(JointJerkTermInfo(coeffs=[0.5] * 7, targets=[0.0] * 7, first_step=0, last_step=29, name="jerk"))
x0 = configs.seeds_for(1, pci, s, g, 256)
ref = None
for lib in ("trajopt_amd/_build_prev/libtrajopt_mi355x.so", "trajopt_amd/_build/libtrajopt_mi355x.so"):
    ctx = runtime.Context(0, lib)
    ctx.upload(pci.to_desc(), abi.default_sqp_params(), abi.default_osqp_settings())
    ctx.set_x0(x0); t0 = time.perf_counter(); ctx.run(0); dt = time.perf_counter() - t0
    r, c = ctx.results(), ctx.counters()
    sig = (r["status"].tobytes(), r["n_qp_solves"].tobytes(), r["x"].tobytes())
    print(lib.split("/")[1], f"{dt*1e3:9.1f} ms", c["n_qp_solves"], f"{c['n_qp_solves']/dt:8.0f} QP/s", "ref" if ref is None else ("bit-identical" if sig == ref else "DIFFERENT"), flush=True)
    ref = ref or sig
    ctx.close()
PY
timeout 600 python -m pytest tests -m gpu -q -x -k "36 or 37" > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2
