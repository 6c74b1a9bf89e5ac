"""Phase profile (needs a -DTMX_PROFILE build: make -C trajopt_amd/csrc EXTRA=-DTMX_PROFILE).
   python tools/prof_phases.py [B] [full|first] [lib.so] [config[s]]   - first QP solve only (default) or the whole optimize() run
   ("full"); config 1 (default), 2, 3 or 4"""
import sys, os, ctypes as C, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trajopt_amd import configs, abi, runtime
smooth = len(sys.argv) > 4 and sys.argv[4].endswith("s")  # "1s": config 1 + acceleration and jerk smoothing costs (banded path)
cid = int(sys.argv[4].rstrip("s")) if len(sys.argv) > 4 else 1
pci, s, g = {1: configs.config1, 2: configs.config2, 3: configs.config3, 4: configs.config4}[cid]()
if smooth:
    from trajopt_amd.problem import JointAccTermInfo, JointJerkTermInfo
    nj = len(pci.cost_infos[0].coeffs) if hasattr(pci.cost_infos[0], "coeffs") else 7
    pci.cost_infos.append(JointAccTermInfo(coeffs=[1.0] * nj, targets=[0.0] * nj, first_step=0, last_step=pci.basic_info.n_steps - 1, name="acc"))
    pci.cost_infos.append(JointJerkTermInfo(coeffs=[0.5] * nj, targets=[0.0] * nj, first_step=0, last_step=pci.basic_info.n_steps - 1, name="jerk"))
desc = pci.to_desc()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
full = len(sys.argv) > 2 and sys.argv[2] == "full"
x0 = configs.seeds_for(cid, pci, s, g, B, **({"sigma": 0.05} if cid in (3, 4) else {}))
ctx = runtime.Context(0, sys.argv[3] if len(sys.argv) > 3 else None)
ctx.upload(desc, abi.default_sqp_params(), configs.osqp_settings_config4() if cid == 4 else abi.default_osqp_settings())
ctx.set_x0(x0)
ctx.kernel_stats(reset=True)
if full:
    ctx.run(0)
    c = ctx.counters()
    iters, nqp = c["admm_iters"], c["n_qp_solves"]
else:
    ctx.convexify()
    ctx.kernel_stats(reset=True)
    xq, cvx, rec = ctx.qp_solve()
    iters, nqp = sum(rec[b].osqp_iter for b in range(B)), B
st = ctx.kernel_stats()
out = (C.c_longlong * 16)()
ctx.lib.tmx_debug_phase_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
ctx.lib.tmx_debug_phase_cycles(ctx.h, out)
names = ["setup", "WALL(10ns)", "ADMM loop | generic: phase A", "check_term | generic: phase B", "convexify_terms | generic: chain", "generic: phase C", "residuals+rho", "polish", "burst entry", "burst exit",
         "store", "qp_structure", "eval+update", "f:assemble", "f:G inverses", "f:Schur+Zs"]
out = list(out)
# Slot 5 is a phase TIME only on the generic path ("phase C").  On the dense fast path (config 1 without smoothing costs) the
# epoch-resident burst uses it as a COUNTER (tmx_part.h: pc[5] += 1 + (go_on << 20) per in-register check): decode it and keep it
# out of the cycle totals - rounds 2 / 3 printed it as 1.18 G "cycles" per problem (83 % of a total that then meant nothing).
fast_path = (cid == 1 and not smooth and not ctx.workspace_in_hbm())
if fast_path:
    # (summed over the batch the low field overflows into the high one: every continued check is a check, so the smallest
    #  carry count k with checks >= continued recovers both)
    n_cont = out[5] >> 20
    n_checks = out[5] - (n_cont << 20)
    while n_checks < n_cont:
        n_checks += 1 << 20
        n_cont -= 1
    print(f"in-register checks of the epoch-resident burst: {n_checks / B:.1f} per problem, {n_cont / B:.1f} of them continued without leaving the burst")
    out[5] = 0
tot = sum(out)
print("B", B, "kernel ms", st["admm_ms"], "admm iters", iters, "qp solves", nqp, "iters/qp", iters / nqp)
for n, c in zip(names, out):
    if c:
        print(f"  {n:12s} {c / B:14.0f} cycles/problem  {100.0 * c / tot:5.1f}%   per-iter {c / max(1, iters):9.1f}   per-qp {c / nqp:10.0f}")
wall = out[1]; tot -= wall
# in-step wall time (100 MHz constant-rate counter) against the sum of the phase cycles (shader clock): their ratio is the
# effective shader clock, and the rows above sum to wall x clock by construction of the ticks
print("  wall-clock per problem (ms)", wall / B * 1e-5, " => effective shader clock (GHz)", tot / max(1, wall) / 10.0)
print("  total cycles/problem", tot / B, " => per ADMM iteration (all phases)", tot / iters)
if hasattr(ctx.lib, "tmx_debug_pspk"):
    # segmented sweeps of the dense-coupling chain (pair-row problems): thread-0 cycles of their three parts, per sweep
    sp = (C.c_longlong * 8)()
    ctx.lib.tmx_debug_pspk.argtypes = [C.POINTER(C.c_longlong)]
    if ctx.lib.tmx_debug_pspk(sp) == 0 and sp[3] > 0:
        n = sp[3]
        print(f"  segmented sweeps: {n} sweeps; cycles per sweep: local sweeps {sp[0] / n:.0f}, boundary vectors {sp[1] / n:.0f}, spike correction {sp[2] / n:.0f}"
              f"  (two sweeps per ADMM iteration: {2 * (sp[0] + sp[1] + sp[2]) / n:.0f} cycles)")
