"""Scaling of the CPU oracle (restated reference path) on the host: threads x processes.  python tools/cpu_scaling.py
Used to pick the fairest cpu_baseline configuration for bench.py (the oracle allocates per term like the reference does)."""
import os, sys, time, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

def worker(nthr, nsample, first):
    from trajopt_amd import configs
    from oracle import pyorc
    pci, s, g = configs.config1()
    x0 = configs.seeds_for(1, pci, s, g, nsample, first=first)
    t0 = time.perf_counter()
    o = pyorc.sqp_batch(pci.to_desc(), x0, nthreads=nthr)
    dt = time.perf_counter() - t0
    print(json.dumps({"iters": int((o["n_func_evals"] - 1).sum()), "dt": dt}))

if len(sys.argv) > 1 and sys.argv[1] == "worker":
    worker(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
    sys.exit(0)
cores = os.cpu_count()
print("host threads", cores)
for nproc, nthr in [(1, 16), (1, 64), (1, cores), (4, cores // 4), (8, cores // 8), (16, cores // 16)]:
    if nthr < 1:
        continue
    nsample = 2 * nthr
    t0 = time.perf_counter()
    ps = [subprocess.Popen([sys.executable, __file__, "worker", str(nthr), str(nsample), str(k * nsample)], stdout=subprocess.PIPE, text=True,
                           env=dict(os.environ, OMP_PROC_BIND="false")) for k in range(nproc)]
    outs = [json.loads(p.communicate()[0].strip().splitlines()[-1]) for p in ps]
    wall = time.perf_counter() - t0
    it = sum(o["iters"] for o in outs)
    print(f"{nproc:3d} proc x {nthr:3d} thr: {it / max(o['dt'] for o in outs):8.0f} SQP it/s (solve time), wall incl. start-up {wall:.1f} s", flush=True)
