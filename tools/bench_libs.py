"""Whole-job throughput of library builds on the bench workload (config 1, B seeds): python tools/bench_libs.py [B] lib.so ...
Prints SQP iters/s (best and median of 3 timed runs after one warm-up) per library; parity-neutral build experiments only."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trajopt_amd import configs, abi, runtime
args = sys.argv[1:]
B = int(args.pop(0)) if args and args[0].isdigit() else 1024
pci, s, g = configs.config1()
desc = pci.to_desc()
x0 = configs.seeds_for(1, pci, s, g, B)
ref = None
for lib in (args or [None]):
    ctx = runtime.Context(0, lib)
    ctx.upload(desc, abi.default_sqp_params(), abi.default_osqp_settings())
    rates = []
    for rep in range(4):
        ctx.set_x0(x0)
        t0 = time.perf_counter(); ctx.run(0); dt = time.perf_counter() - t0
        r = ctx.results()
        iters = int((r["n_func_evals"] - 1).sum()) if "n_func_evals" in r else 0
        if rep:
            rates.append(r["n_qp_solves"].sum() / dt)
    sig = (r["status"].tobytes(), r["n_qp_solves"].tobytes(), r["x"].tobytes())
    same = "ref" if ref is None else ("bit-identical" if sig == ref else "DIFFERENT RESULTS")
    ref = ref or sig
    name = os.path.basename(os.path.dirname(lib)) if lib else "default"
    print(f"{name:16s} B={B}: best {max(rates):9.0f}  median {sorted(rates)[1]:9.0f} QP solves/s   [{same}]", flush=True)
    ctx.close()
