"""Per-wave phase profile of the ADMM iteration of k_sqp_pool (config 1): needs a build with -DTMX_PROFILE -DTMX_PROFILE_LOOP=2
[-DTMX_PROF_TID=<thread>] (tools/build_variants.sh).   python tools/prof_loop.py [B] lib.so [lib.so ...]
Prints, per ADMM iteration and as seen by thread TMX_PROF_TID, the cycles of every phase's work and of the wait at the barrier that
ends it (the eleven TMX_LT points of admm_burst_core's iteration)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajopt_amd import configs, abi, runtime
args = sys.argv[1:]
B = int(args.pop(0)) if args and args[0].isdigit() else 256
pci, s, g = configs.config1()
desc = pci.to_desc()
x0 = configs.seeds_for(1, pci, s, g, B)
names = {0: "C tail + A work", 2: "  barrier 1 wait", 3: "B (A'e gather) work", 5: "  barrier 2 wait", 6: "interior work", 7: "  barrier 3 wait",
         8: "separator work", 9: "  barrier 4 wait", 10: "correction work", 13: "  barrier 5 wait", 14: "C work"}
for lib in args:
    ctx = runtime.Context(0, lib)
    ctx.upload(desc, abi.default_sqp_params(), abi.default_osqp_settings())
    ctx.set_x0(x0)
    ctx.kernel_stats(reset=True)
    ctx.run(0)
    iters = ctx.counters()["admm_iters"]
    out = (C.c_longlong * 16)()
    ctx.lib.tmx_debug_phase_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
    ctx.lib.tmx_debug_phase_cycles(ctx.h, out)
    out = list(out)
    tot = sum(out[k] for k in names)
    print(f"{os.path.basename(os.path.dirname(lib))}: B {B}, {iters} ADMM iterations, loop cycles per iteration {tot / iters:.0f}")
    for k in sorted(names):
        print(f"   {names[k]:22s} {out[k] / iters:8.1f}")
    ctx.close()
