#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on BASELINE.json's config.

A "step" is one complete pass of the hot path over one batch of synthetic input: BasicTrustRegionSQP::optimize()
(convexify + batched QP solves + exact re-evaluation + trust-region/penalty logic) for 1024 random seeds per GPU
of the glass_upright problem (config 1: 7-DOF, 30 waypoints, JointVel cost, upright CartPose constraints, goal
JointPos constraint, single-timestep collision cost).  Seeds are resident in HBM before the timed region.
  value = SQP iterations / s over all GPUs  (one SQP iteration = one trust-region evaluation = one QP solve + one
          exact re-evaluation; the reference's counters n_func_evals-1 / n_qp_solves, optimizers.hpp:47)
N > 1: one process per GPU (torch.distributed, backend nccl = RCCL, for the launcher's barrier / timing reductions), each rank
owns its own 1024 seeds (weak scaling, no data-path collective); the only exchange on the path is the best-seed reduction
after every step, done by the library itself over its own RCCL communicator (tmx_nccl_init / tmx_argmin: an all-gather of
one 16-byte (cost, index) pair per rank).  The communicator is created at N = 1 too, so the collective is exercised by
every bench run.
`--config 2|3|4` benches the other BASELINE configurations (puzzle_piece, car_seat, trajopt_sqp path) with their own
roofline object; the default and the headline is config 1.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6  # MI355X fp64 vector = matrix peak (spec) — the guide lists no fp64 MFMA row
BATCH_PER_GPU = 1024


def algorithmic_flops_per_admm_iter(T, D, n, m, nnzA):
    """SURVEY.md §8(d) Phase B: 2 block-triangular sweeps (T*8*D^2) + 2 SpMV with A (4*nnz(A)) + ~12*(n+m) vector ops"""
    return 8.0 * T * D * D + 4.0 * nnzA + 12.0 * (n + m)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0, help="seeds per GPU (default: the configuration's BASELINE batch per GPU)")
    ap.add_argument("--depth", type=int, default=None, help="batches in flight (1 = strictly sequential steps, 2 = double-buffered contexts); default 2, and 1 for the HBM-workspace configurations 2 / 3, whose launches are enqueued at once: with two in flight their HIP start event is recorded while the kernel still queues behind the other context, so the launch time would include queue wait and roofline.frac would not be a kernel figure")
    ap.add_argument("--config", type=int, default=1, choices=[1, 2, 3, 4], help="BASELINE.json configuration (default 1 = the metric's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from trajopt_amd import abi, configs, runtime

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    cid = args.config
    # (constructor, BASELINE batch per GPU, seed sigma, workload text)
    spec = {1: (configs.config1, 1024, 0.1, "config 1: glass_upright 7-DOF 30-waypoint, collision cost, batch=%d random seeds per GPU"),
            2: (configs.config2, 256, None, "config 2: puzzle_piece 7-DOF 300-waypoint, 6-row CartPose constraint per waypoint, batch=%d per GPU"),
            3: (configs.config3, 128, 0.05, "config 3: car_seat 10-DOF 50-waypoint, 20 obstacles, LVS_CONTINUOUS collision, batch=%d per GPU "
                                            "(BASELINE: 512 over 4 GPUs)"),
            4: (configs.config4, 1024, 0.05, "config 4: trajopt_ifopt/trajopt_sqp path 7-DOF 30-waypoint, continuous collision hinge cost, "
                                             "batch=%d problems per GPU (BASELINE: 8192 over 8 GPUs)")}[cid]
    pci, start, goal = spec[0]()
    desc = pci.to_desc()
    T, D = pci.basic_info.n_steps, pci.robot.n_dof
    B = args.batch or spec[1]
    conv_code = abi.SQP_CONVERGED if cid == 4 else abi.OPT_CONVERGED
    osqp_st = configs.osqp_settings_config4() if cid == 4 else abi.default_osqp_settings()
    # Two contexts = two batches in flight (double buffering): the kernel time of a batch is set by its longest chain of QP
    # solves (one straggler seed with 104 instead of 61 solves stretches the launch from 475 to 604 ms, tools/tail_probe.py),
    # the persistent workgroups retire when nothing is left for them, and the next batch's kernel - enqueued on the other
    # context's stream - takes over their CUs.  Every step still is one complete optimize() of one batch; all of them start
    # and end inside the timed region.  --depth 1 runs them strictly one after the other.
    depth = max(1, min(2, args.depth if args.depth is not None else (1 if cid in (2, 3) else 2)))
    ctxs = []
    for _ in range(depth):
        c = runtime.Context(local_rank)
        c.upload(desc, abi.default_sqp_params(), osqp_st)
        # the library's own RCCL communicator for the best-seed reduction (also with one rank: the collective always runs)
        uid = torch.zeros(128, dtype=torch.uint8, device=dev)
        if rank == 0:
            uid = torch.tensor(list(c.nccl_unique_id()), dtype=torch.uint8, device=dev)
        if world > 1:
            dist.broadcast(uid, src=0)
        c.nccl_init(bytes(uid.cpu().tolist()), world, rank)
        ctxs.append(c)
    ctx = ctxs[0]

    nsteps = args.warmup + args.steps
    # synthetic seeds (counter-based Philox keyed by (config, global problem index)), resident in HBM before timing
    kw = {} if spec[2] is None else {"sigma": spec[2]}
    seeds_host = configs.seeds_for(cid, pci, start, goal, B * nsteps, first=rank * B * nsteps, **kw)
    seeds = torch.from_numpy(seeds_host.reshape(nsteps, B, T, D)).to(dev)
    torch.cuda.synchronize()

    def issue(k):
        c = ctxs[k % depth]
        c.set_x0_device(seeds[k].data_ptr(), B)
        c.launch()

    def finish(k):
        c = ctxs[k % depth]
        n_active = c.wait()
        assert n_active == 0, f"step {k}: {n_active} problems left unfinished by a run-to-completion launch"
        r = c.results()
        best = c.argmin((rank * nsteps + k) * B)   # the only collective: RCCL all-gather of 16 bytes per rank inside the library
        if best[0] >= 0:
            c.best_trajectory()   # ... and its optional last step: the winning T x D trajectory broadcast from its owner rank
        return r, best, c.counters()

    for k in range(args.warmup):   # warm-up steps run one at a time
        issue(k)
        finish(k)
    for c in ctxs:
        c.kernel_stats(reset=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tot_fe = tot_qp = tot_admm = 0
    conv = 0
    last_r, last_best, last_k = None, None, None
    for k in range(args.warmup, min(nsteps, args.warmup + depth - 1)):
        issue(k)
    for k in range(args.warmup, nsteps):
        if k + depth - 1 < nsteps:
            if depth > 1:
                # the next batch is enqueued when this one begins to retire workgroups (a pinned word the kernel sets): its
                # launch then lasts from about its first to its last workgroup, as an isolated launch does, instead of
                # sitting in the queue behind this batch's bulk with its start event already recorded
                cur = ctxs[k % depth]
                t_poll = time.perf_counter()
                while not cur.tail_started() and time.perf_counter() - t_poll < 120.0:
                    time.sleep(0.0002)
            issue(k + depth - 1)
        r, best, c = finish(k)
        last_r, last_best, last_k = r, best, k
        # trajopt_sqp counts QP solves only (SQPResults::overall_iteration): one trust-region evaluation each
        tot_fe += int(r["n_qp_solves"].sum()) if cid == 4 else int((r["n_func_evals"] - 1).sum())
        tot_qp += int(r["n_qp_solves"].sum())
        tot_admm += c["admm_iters"]
        conv += int((r["status"] == conv_code).sum())
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    stats = {"admm_ms": 0.0, "convexify_ms": 0.0, "evaluate_ms": 0.0, "admm_launches": 0}
    for c in ctxs:
        st_c = c.kernel_stats()
        for key in stats:
            stats[key] += st_c[key]
    tt = torch.tensor([elapsed, float(tot_fe), float(tot_qp), float(tot_admm), float(conv)], dtype=torch.float64, device=dev)
    if world > 1:
        tmax = tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tt.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        g_fe, g_qp, g_admm, g_conv = (float(tsum[i]) for i in range(1, 5))
    else:
        g_fe, g_qp, g_admm, g_conv = float(tot_fe), float(tot_qp), float(tot_admm), float(conv)

    if rank == 0:
        # ---- roofline of the dominant kernel, rank 0, HIP-event timed on the library's own stream ----
        # optimize() is ONE persistent launch of k_sqp_pool per step (convexify + QP solves + exact re-evaluation + SQP
        # decisions for every seed); >90 % of it is the batched OSQP-style ADMM.  The arithmetic is fp64 FMA on the
        # vector ALUs; on MI355X the fp64 vector peak equals the fp64 matrix peak (78.6 TFLOP/s), which is the "mfma"
        # roof the contract asks for.  The kernel is bound by dependent-issue latency, not by that roof.
        recs, cnt = ctx.qp_records(4)
        r0 = recs[0]
        f_iter = algorithmic_flops_per_admm_iter(T, D, r0.n, r0.m, r0.nnzA)
        launches = max(1, stats["admm_launches"])
        # HBM bytes per launch from the rocprofv3 PMC passes of this same command (profiles/README.md); null if the
        # summary is not present
        traffic = None
        traffic_source = None
        mfma_ops = None
        try:
            import glob
            pat = "r*_pmc_traffic.json" if cid == 1 else "r*_pmc_traffic_cfg%d.json" % cid
            tfile = sorted(glob.glob(os.path.join(ROOT, "profiles", pat)))[-1]   # latest round
            with open(tfile) as f:
                tj = json.load(f)
            if tj.get("batch_per_gpu") == B:
                traffic = tj.get("hbm_bytes_per_launch")
                # the counters are NOT collected in this run (rocprofv3 --pmc needs its own passes, tools/pmc_traffic.sh): the number is
                # read from the committed summary of the latest collection of this same command
                traffic_source = "committed counter summary profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command " \
                                 "on an earlier run; not measured in this run)" % os.path.basename(tfile)
                mfma_ops = tj.get("mfma_mops_f64_per_launch")   # SQ_INSTS_VALU_MFMA_MOPS_F64 of the same collection (None: not collected)
        except (OSError, IndexError):
            pass
        flops_per_launch = f_iter * tot_admm / launches
        avg_ms = stats["admm_ms"] / launches
        achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        roofline = {
            "kernel": "k_sqp_pool", "bound": "mfma", "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / FP64_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_source,
            # "bound": "mfma" is the contract's label for the flop roof (fp64 vector peak = fp64 matrix peak on this part); how many
            # MFMA operations the kernel really issues is this counter (per launch, same collection as `traffic`)
            "mfma_ops": mfma_ops,
            "avg_launch_ms": avg_ms, "launches": launches, "algorithmic_flop_per_admm_iter": f_iter,
            "admm_iters_per_launch": tot_admm / launches,
            # with two batches in flight the launches overlap (the tail of one under the bulk of the next): a launch still lasts
            # avg_launch_ms from its first to its last workgroup, but one completes every ms_per_step
            "batches_in_flight": depth,
            "achieved_at_step_rate": flops_per_launch / (elapsed / max(1, args.steps)) / 1e12,
            "kernel_time_share": {"admm_ms": stats["admm_ms"], "convexify_ms": stats["convexify_ms"],
                                  "evaluate_ms": stats["evaluate_ms"], "wall_ms": (t1 - t0) * 1e3},
        }
        kernel_name = "k_sqp_fused_hbm" if ctx.workspace_in_hbm() else ("k_sqp_wave" if (cid == 1 and os.environ.get("TMX_WAVE") == "1") else "k_sqp_pool")
        roofline["kernel"] = kernel_name
        if kernel_name == "k_sqp_fused_hbm":
            # QP workspace in HBM (configs 2 and 3): the rows and the block factor are streamed from the workgroup's HBM slice every
            # ADMM iteration - SURVEY.md section 8(d): B_admm = 2 * 12 * (nnz(L) + nnz(A)) bytes per iteration,
            # nnz(L) ~ T * 1.5 * D^2 + the slack couplings (one per aux variable).  `traffic` is the counter measurement of the same
            # launch (profiles/r*_pmc_traffic_cfgN.json): what the kernel really moved, against these algorithmic bytes.
            nnzL = 1.5 * T * D * D + (r0.n - T * D)
            b_iter = 24.0 * (nnzL + r0.nnzA)
            ach = b_iter * tot_admm / launches / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            roofline = {"kernel": kernel_name, "bound": "hbm", "achieved": ach, "peak": 8000.0,
                        "unit": "GB/s", "frac": ach / 8000.0, "traffic": traffic, "traffic_source": traffic_source,
                        "mfma_ops": mfma_ops, "avg_launch_ms": avg_ms, "launches": launches,
                        "algorithmic_bytes_per_admm_iter": b_iter, "admm_iters_per_launch": tot_admm / launches,
                        "measured_traffic_gbps": (traffic / (avg_ms * 1e-3) / 1e9) if (traffic and avg_ms > 0) else None,
                        "flop_view": {"achieved_tflops": achieved, "frac_of_fp64_peak": achieved / FP64_PEAK_TFLOPS}}
        # (config 4 runs LDS-resident on k_sqp_pool's generic path: like config 1 it is priced against the fp64 roof - it is
        #  bound by dependent-issue latency, not by HBM)
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            from oracle import pyorc
            cores = os.cpu_count() or 1
            pyorc.build()
            # one problem per OpenMP thread.  Several thread counts are timed and the FASTEST is reported, so the GPU/CPU
            # ratio is not flattered by a badly subscribed baseline.  (Round-2 probe on the GPU box, tools/cpu_scaling.py:
            # 16 threads 1.57 k, 64 threads 1.64 k, 256 threads 0.84 k SQP it/s, and 4 x 64 / 8 x 32 / 16 x 16 processes x
            # threads 0.84-0.86 k: the box gives the container about 16 cores' worth of CPU time whatever the thread or
            # process count - it is a CPU quota, not the oracle's allocator, that bounds this leg.)
            quota = None
            try:
                with open("/sys/fs/cgroup/cpu.max") as f:
                    q = f.read().split()
                if q[0] != "max":
                    quota = float(q[0]) / float(q[1])
            except (OSError, ValueError, IndexError):
                pass
            tried = []
            # Bounded (~15 - 25 s): a short probe picks the thread count, then THREE repetitions of >= 3.5 s each on fresh seeds of the
            # workload at that count; `value` is their median, `min` / `max` their spread (VERDICT of round 5: a 1.5 s sample moved by
            # 20 % between runs).  (Round-2 probe on the GPU box, tools/cpu_scaling.py: 16 threads 1.57 k, 64 threads 1.64 k, 256 threads
            # 0.84 k SQP it/s: the box gives the container about 16 cores' worth of CPU time - a quota, reported below.)
            def cpu_sample(nthr, min_s, lo0):
                nsample = min(B, max(32, 2 * nthr)) if cid == 1 else min(B, max(64, 2 * nthr))
                nit, nqp, nadmm, dt, done = 0.0, 0.0, 0.0, 0.0, 0
                while True:
                    lo = (lo0 + done) % max(1, len(seeds_host) - nsample + 1)
                    xs = seeds_host[lo:lo + nsample]
                    tc0 = time.perf_counter()
                    o = pyorc.sqp2_batch(desc, xs, osqp=osqp_st, nthreads=nthr) if cid == 4 else pyorc.sqp_batch(desc, xs, nthreads=nthr)
                    dt += time.perf_counter() - tc0
                    nit += float(o["n_qp_solves"].sum() if cid == 4 else (o["n_func_evals"] - 1).sum())
                    nqp += float(o["n_qp_solves"].sum())
                    nadmm += float(o["admm_iters"]) if "admm_iters" in o else float("nan")
                    done += nsample
                    if dt >= min_s or dt > 30.0:
                        break
                return (float(nit / dt), nthr, done, dt, nqp, nadmm)
            thread_counts = sorted({min(cores, 64), min(cores, 32), min(cores, 16)}, reverse=True) if cid == 1 else [min(cores, 32), min(cores, 16)]
            for nthr in thread_counts:
                tried.append(cpu_sample(nthr, 0.8 if cid == 1 else 1.5, 0))
            nbest = max(tried, key=lambda t: t[0])[1]
            reps = [cpu_sample(nbest, 3.5, 64 * (i + 1)) for i in range(3)]
            vals = sorted(t[0] for t in reps)
            best = sorted(reps, key=lambda t: t[0])[1]   # the median repetition
            cpu = {"value": vals[1], "unit": "SQP iters/s", "cores": best[1], "kind": "port",
                   "min": vals[0], "max": vals[2], "repetitions": 3,
                   "sample": f"3 x >= 3.5 s ({sum(t[2] for t in reps)} seeds of the same workload in all, {sum(t[3] for t in reps):.1f} s wall), one problem per "
                             f"OpenMP thread on {best[1]} of {cores} hardware threads; median of the three, min / max beside it; restated reference "
                             "CPU path (oracle/), not the upstream binary; thread-count probe: " +
                             ", ".join(f"{t[1]} thr -> {t[0]:.0f} it/s" for t in tried),
                   "per_thread_value": vals[1] / best[1], "host_threads": cores, "cgroup_cpu_quota_cores": quota,
                   "qp_solves_per_s": best[4] / best[3],
                   "admm_iters_per_s": (best[5] / best[3] if best[5] == best[5] else None)}
        # parity of THIS run's results (after the timed region; the oracle is the checker, never the thing measured): the first 64 problems
        # of the last timed batch against the restated reference CPU path on the same seeds - north_star: "joint trajectories within
        # 1e-5 rad".  (SQP outcomes are discontinuous in rounding: the seeds outside the ball are the ones on which the oracle parts from
        # its own FMA build as well, tests/test_gpu_parity.py.)
        parity = None
        if world == 1 and not args.no_cpu_baseline and last_r is not None:
            from oracle import pyorc
            npar = min(64, B)
            k_last = nsteps - 1
            xs = seeds_host[k_last * B:k_last * B + npar]
            o = pyorc.sqp2_batch(desc, xs, osqp=osqp_st, nthreads=min(os.cpu_count() or 1, 32)) if cid == 4 else \
                pyorc.sqp_batch(desc, xs, nthreads=min(os.cpu_count() or 1, 32))
            dxp = np.abs(last_r["x"][:npar] - o["x"]).reshape(npar, -1).max(axis=1)
            parity = {"seeds_compared": int(npar), "parity_frac_within_1e-5": float((dxp <= 1e-5).mean()),
                      "same_status_frac": float((last_r["status"][:npar] == o["status"]).mean()),
                      "same_n_qp_solves_frac": float((last_r["n_qp_solves"][:npar] == o["n_qp_solves"]).mean()),
                      "median_abs_dx": float(np.median(dxp)), "max_abs_dx": float(dxp.max()),
                      "what": "first %d problems of the last timed batch vs the oracle (oracle/, restated reference CPU path) on the same seeds" % npar}
            # What the CONSUMER of a multi-seed run sees (VERDICT of round 5): tmx_argmin's winner of the last batch - index and cost -
            # against the oracle.  The oracle cannot run all B seeds inside a bench (B x ~1.7 s of CPU): it runs the 32 seeds the device
            # ranks best, and must (i) rank the same seed first among them, (ii) reproduce its cost; the full comparison (oracle on every
            # seed of a 256-seed batch) is tests/test_gpu_parity.py::test_argmin_of_a_batch_matches_the_oracle.
            if cid != 4 and last_best is not None and last_best[0] >= 0:
                off = (rank * nsteps + last_k) * B
                cost = np.where(last_r["status"] == conv_code, last_r["total_cost"], np.inf)
                top = np.argsort(cost, kind="stable")[:32]
                ot = pyorc.sqp_batch(desc, seeds_host[last_k * B + top], nthreads=min(os.cpu_count() or 1, 32))
                ocost = np.where(ot["status"] == conv_code, ot["total_cost"], np.inf)
                ow = int(top[int(np.argmin(ocost))])
                parity["argmin"] = {"device_index": int(last_best[0] - off), "device_cost": float(last_best[1]),
                                    "oracle_index_among_the_devices_best_32": ow, "oracle_cost": float(ocost.min()),
                                    "same_winner": bool(ow == int(last_best[0] - off)),
                                    "abs_cost_diff": float(abs(ocost.min() - last_best[1])),
                                    "max_abs_cost_diff_over_the_32": float(np.abs(ocost - cost[top])[np.isfinite(ocost) & np.isfinite(cost[top])].max(initial=0.0))}
        line = {
            "metric": "SQP iters/s (+ QP solves/s), 7-DOF x 30-wp x 1024-batch glass_upright" if cid == 1 else
                      "SQP iters/s (+ QP solves/s), BASELINE config %d" % cid,
            "value": g_fe / elapsed, "unit": "SQP iters/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / max(1, args.steps) * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "batches_in_flight": depth,
            "config": {"workload": spec[3] % B,
                       "n_dof": D, "n_steps": T, "batch_per_gpu": B, "qp_n": r0.n, "qp_m": r0.m, "parallelism": "seeds sharded, dp%d" % world},
            "qp_solves_per_s": g_qp / elapsed, "admm_iters_per_s": g_admm / elapsed,
            "converged_frac": g_conv / (B * world * max(1, args.steps)),
            "roofline": roofline, "cpu_baseline": cpu, "parity": parity,
        }
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for c in ctxs:
        c.close()


if __name__ == "__main__":
    main()
