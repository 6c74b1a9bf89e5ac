"""ctypes binding of libtrajopt_mi355x.so + the host-side mirror of the reference optimizer surface.

`BatchedTrustRegionSQP` mirrors sco::BasicTrustRegionSQP (trajopt_sco/include/trajopt_sco/optimizers.hpp:137-218:
setParameters / initialize / optimize / results) for a BATCH of seeds of one trajopt problem; everything numerical
happens in the HIP kernels behind the C-ABI (include/tmx.h).  There is NO CPU fallback: constructing a Context
without a HIP device raises.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_DEFAULT_LIB = os.path.join(_HERE, "_build", "libtrajopt_mi355x.so")

_SIGS = {
    "tmx_create": ([C.c_int, C.POINTER(C.c_void_p)], C.c_int),
    "tmx_destroy": ([C.c_void_p], None),
    "tmx_last_error": ([C.c_void_p], C.c_char_p),
    "tmx_default_sqp_params": ([C.POINTER(abi.SqpParams)], None),
    "tmx_default_osqp_settings": ([C.POINTER(abi.OsqpSettings)], None),
    "tmx_problem_upload": ([C.c_void_p, C.POINTER(abi.ProblemDesc), C.POINTER(abi.SqpParams), C.POINTER(abi.OsqpSettings)], C.c_int),
    "tmx_batch_set_x0": ([C.c_void_p, C.c_void_p, C.c_int32], C.c_int),
    "tmx_batch_set_x0_device": ([C.c_void_p, C.c_void_p, C.c_int32], C.c_int),
    "tmx_sqp_set_x": ([C.c_void_p, C.c_void_p], C.c_int),
    "tmx_sqp_stop": ([C.c_void_p, C.c_int32, C.c_int32], C.c_int),
    "tmx_best_trajectory": ([C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)], C.c_int),
    "tmx_sqp_run": ([C.c_void_p, C.c_int32, C.POINTER(C.c_int32)], C.c_int),
    "tmx_sqp_launch": ([C.c_void_p], C.c_int),
    "tmx_sqp_wait": ([C.c_void_p, C.POINTER(C.c_int32)], C.c_int),
    "tmx_sqp_tail_started": ([C.c_void_p], C.c_int32),
    "tmx_sqp_results": ([C.c_void_p] + [C.c_void_p] * 5, C.c_int),
    "tmx_sqp_counters": ([C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)], C.c_int),
    "tmx_sqp_qp_records": ([C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p], C.c_int),
    "tmx_term_counts": ([C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)], C.c_int),
    "tmx_evaluate": ([C.c_void_p, C.c_void_p, C.c_void_p], C.c_int),
    "tmx_convexify": ([C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p], C.c_int),
    "tmx_export_csc": ([C.c_void_p, C.c_int32] + [C.POINTER(C.c_int32)] * 4 + [C.c_void_p] * 9, C.c_int),
    "tmx_qp_dims": ([C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)], C.c_int),
    "tmx_qp_solve": ([C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p], C.c_int),
    "tmx_qp_solve_batched": ([C.c_void_p, C.POINTER(abi.QpCsc), C.c_int32, C.POINTER(abi.OsqpSettings), C.c_void_p, C.c_void_p,
                              C.c_void_p, C.POINTER(abi.QpInfo), C.c_void_p], C.c_int),
    "tmx_qp_active_set": ([C.c_void_p, C.c_void_p], C.c_int),
    "tmx_qp_duals": ([C.c_void_p, C.c_void_p], C.c_int),
    "tmx_argmin": ([C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_double)], C.c_int),
    "tmx_attach_nccl": ([C.c_void_p, C.c_void_p], C.c_int),
    "tmx_nccl_unique_id": ([C.c_void_p], C.c_int),
    "tmx_nccl_init": ([C.c_void_p, C.c_void_p, C.c_int32, C.c_int32], C.c_int),
    "tmx_kernel_stats": ([C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double)], C.c_int),
    "tmx_kernel_stats_reset": ([C.c_void_p], C.c_int),
    "tmx_sqp_state": ([C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p], C.c_int),
    "tmx_sqp_step_log": ([C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)], C.c_int),
    "tmx_workspace_info": ([C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64)], C.c_int),
    "tmx_model_values": ([C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p], C.c_int),
    "tmx_sqp_set_loop_vars": ([C.c_void_p, C.c_void_p, C.c_void_p], C.c_int),
}
ABI_SYMBOLS = tuple(_SIGS.keys())


class TmxError(RuntimeError):
    pass


def load_library(path: str = None):
    path = path or _DEFAULT_LIB
    if not os.path.exists(path):
        raise TmxError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(the HIP extension is required; there is no CPU fallback)")
    lib = C.CDLL(path)
    for name, (argt, rest) in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = argt
        fn.restype = rest
    return lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Context:
    """one tmx_ctx (one GPU, one host thread)"""

    def __init__(self, device: int = 0, lib_path: str = None):
        self.lib = load_library(lib_path)
        h = C.c_void_p()
        rc = self.lib.tmx_create(device, C.byref(h))
        if rc != abi.TMX_OK:
            raise TmxError(f"tmx_create failed with status {rc}: no usable HIP device (the product path needs an MI355X)")
        self.h = h
        self.desc = None
        self.B = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.tmx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != abi.TMX_OK:
            raise TmxError(f"tmx status {rc}: {self.lib.tmx_last_error(self.h).decode()}")

    # ---- S4 ----
    def upload(self, desc: abi.ProblemDesc, sqp: abi.SqpParams = None, osqp: abi.OsqpSettings = None):
        self.desc = desc
        self._chk(self.lib.tmx_problem_upload(self.h, C.byref(desc), C.byref(sqp) if sqp is not None else None,
                                              C.byref(osqp) if osqp is not None else None))
        nc, nn, nr = C.c_int32(), C.c_int32(), C.c_int32()
        self._chk(self.lib.tmx_term_counts(self.h, C.byref(nc), C.byref(nn), C.byref(nr)))
        self.n_costs, self.n_cnts, self.R = nc.value, nn.value, nr.value
        nm, mm = C.c_int32(), C.c_int32()
        self._chk(self.lib.tmx_qp_dims(self.h, C.byref(nm), C.byref(mm)))
        self.n_max, self.m_max = nm.value, mm.value
        self.T, self.D = desc.n_steps, desc.n_dof + (1 if desc.use_time else 0)   # columns of a trajectory (joints + the time column)

    # ---- S3 ----
    def set_x0(self, x0):
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        self.B = x0.shape[0]
        assert x0.size == self.B * self.T * self.D
        self._chk(self.lib.tmx_batch_set_x0(self.h, _ptr(x0), self.B))

    def set_x(self, x):
        """QPProblem::setVariables: overwrite the iterate of every problem, keep the convexification and all loop state"""
        x = np.ascontiguousarray(x, dtype=np.float64)
        assert x.size == self.B * self.T * self.D
        self._chk(self.lib.tmx_sqp_set_x(self.h, _ptr(x)))

    def set_x0_device(self, dev_ptr: int, batch: int):
        self.B = batch
        self._chk(self.lib.tmx_batch_set_x0_device(self.h, C.c_void_p(dev_ptr), batch))

    def run(self, max_steps: int = 0) -> int:
        na = C.c_int32(0)
        self._chk(self.lib.tmx_sqp_run(self.h, max_steps, C.byref(na)))
        return na.value

    def launch(self):
        """asynchronous half of run(0): the whole optimize() of the batch is enqueued on the context's stream"""
        self._chk(self.lib.tmx_sqp_launch(self.h))

    def tail_started(self) -> bool:
        """the pending launch has begun to retire workgroups (host-side poll of one pinned word)"""
        return bool(self.lib.tmx_sqp_tail_started(self.h))

    def wait(self) -> int:
        na = C.c_int32(0)
        self._chk(self.lib.tmx_sqp_wait(self.h, C.byref(na)))
        return na.value

    def results(self):
        B = self.B
        x = np.zeros((B, self.T, self.D))
        status = np.zeros(B, np.int32)
        cost = np.zeros(B)
        nfe = np.zeros(B, np.int32)
        nqp = np.zeros(B, np.int32)
        self._chk(self.lib.tmx_sqp_results(self.h, _ptr(x), _ptr(status), _ptr(cost), _ptr(nfe), _ptr(nqp)))
        return dict(x=x, status=status, total_cost=cost, n_func_evals=nfe, n_qp_solves=nqp)

    def stop(self, problem: int, status: int):
        """finish one problem of the batch between bounded run() calls (tmx_sqp_stop: a callback returned false)"""
        self._chk(self.lib.tmx_sqp_stop(self.h, int(problem), int(status)))

    def state(self):
        """loop variables of BasicTrustRegionSQP::optimize per problem (between bounded run() calls)"""
        B = self.B
        it, mi, done = np.zeros(B, np.int32), np.zeros(B, np.int32), np.zeros(B, np.int32)
        trust = np.zeros(B)
        self._chk(self.lib.tmx_sqp_state(self.h, _ptr(it), _ptr(mi), _ptr(trust), _ptr(done)))
        return dict(sqp_iter=it, merit_increases=mi, trust_box_size=trust, done=done.astype(bool))

    def step_log(self):
        """BasicTrustRegionSQPResults of the last trust-region evaluation of every problem (tmx_sqp_step_log): the columns of
        the reference's per-iteration table (optimizers.cpp:428-531).  Returns a list of dicts, one per problem."""
        stride = C.c_int32(0)
        self._chk(self.lib.tmx_sqp_step_log(self.h, None, C.byref(stride)))
        out = np.zeros((self.B, stride.value))
        self._chk(self.lib.tmx_sqp_step_log(self.h, _ptr(out), C.byref(stride)))
        nc, nv, H = self.n_costs, self.n_cnts, abi.STEP_LOG_HEAD
        logs = []
        for b in range(self.B):
            o = out[b]
            q = o[H:]
            logs.append(dict(merit_increases=int(o[0]), sqp_iter=int(o[1]), box_size=float(o[2]), old_merit=float(o[3]), model_merit=float(o[4]),
                             new_merit=float(o[5]), approx_merit_improve=float(o[6]), exact_merit_improve=float(o[7]),
                             merit_improve_ratio=float(o[8]), valid=bool(o[9]),
                             old_cost_vals=q[0:nc].copy(), model_cost_vals=q[nc:2 * nc].copy(), new_cost_vals=q[2 * nc:3 * nc].copy(),
                             old_cnt_viols=q[3 * nc:3 * nc + nv].copy(), model_cnt_viols=q[3 * nc + nv:3 * nc + 2 * nv].copy(),
                             new_cnt_viols=q[3 * nc + 2 * nv:3 * nc + 3 * nv].copy(), merit_error_coeffs=q[3 * nc + 3 * nv:3 * nc + 4 * nv].copy()))
        return logs

    def workspace_info(self):
        a, b, c = C.c_int32(0), C.c_int64(0), C.c_int64(0)
        self._chk(self.lib.tmx_workspace_info(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(in_hbm=bool(a.value), lds_bytes=b.value, hbm_bytes_per_problem=c.value)

    def workspace_in_hbm(self) -> bool:
        return self.workspace_info()["in_hbm"]

    def model_values(self, x_qp):
        """evaluateModelCosts / evaluateModelCntViols (trajopt_sqp: evaluateConvexCosts / evaluateConvexConstraintViolations) of
        the current convexification at QP variables x_qp[B][n_max] (reference order)"""
        x_qp = np.ascontiguousarray(x_qp, np.float64)
        assert x_qp.shape == (self.B, self.n_max)
        mc, mv = np.zeros((self.B, self.n_costs)), np.zeros((self.B, self.n_cnts))
        self._chk(self.lib.tmx_model_values(self.h, _ptr(x_qp), _ptr(mc), _ptr(mv)))
        return mc, mv

    def set_loop_vars(self, trust_box_size=None, merit_error_coeffs=None):
        """trust box size [B] / merit coefficients [B][n_cnts] of an outer optimizer that drives the piecewise hooks itself"""
        t = None if trust_box_size is None else np.ascontiguousarray(np.broadcast_to(np.asarray(trust_box_size, np.float64), (self.B,)))
        m = None if merit_error_coeffs is None else np.ascontiguousarray(np.broadcast_to(np.asarray(merit_error_coeffs, np.float64), (self.B, self.n_cnts)))
        self._chk(self.lib.tmx_sqp_set_loop_vars(self.h, _ptr(t), _ptr(m)))

    def counters(self):
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        self._chk(self.lib.tmx_sqp_counters(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(n_func_evals=a.value, n_qp_solves=b.value, admm_iters=c.value)

    def qp_records(self, max_records: int = 64):
        recs = (abi.QpRecord * (self.B * max_records))()
        cnt = np.zeros(self.B, np.int32)
        self._chk(self.lib.tmx_sqp_qp_records(self.h, recs, max_records, _ptr(cnt)))
        return recs, cnt

    # ---- piecewise hooks ----
    def evaluate(self):
        cv = np.zeros((self.B, self.n_costs))
        vv = np.zeros((self.B, self.n_cnts))
        self._chk(self.lib.tmx_evaluate(self.h, _ptr(cv), _ptr(vv)))
        return cv, vv

    def convexify(self):
        act = np.zeros((self.B, self.R), np.int32)
        coef = np.zeros((self.B, self.R, self.D))
        rhs = np.zeros((self.B, self.R))
        self._chk(self.lib.tmx_convexify(self.h, _ptr(act), _ptr(coef), _ptr(rhs)))
        return act, coef, rhs

    def export_csc(self, problem: int = 0):
        n, m, nzp, nza = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        args = (self.h, problem, C.byref(n), C.byref(m), C.byref(nzp), C.byref(nza))
        self._chk(self.lib.tmx_export_csc(*args, *([None] * 9)))
        Pp, Pi, Px = np.zeros(n.value + 1, np.int64), np.zeros(nzp.value, np.int64), np.zeros(nzp.value)
        Ap, Ai, Ax = np.zeros(n.value + 1, np.int64), np.zeros(nza.value, np.int64), np.zeros(nza.value)
        q, l, u = np.zeros(n.value), np.zeros(m.value), np.zeros(m.value)
        self._chk(self.lib.tmx_export_csc(*args, _ptr(Pp), _ptr(Pi), _ptr(Px), _ptr(q), _ptr(Ap), _ptr(Ai), _ptr(Ax),
                                          _ptr(l), _ptr(u)))
        return dict(n=n.value, m=m.value, P_p=Pp, P_i=Pi, P_x=Px, q=q, A_p=Ap, A_i=Ai, A_x=Ax, l=l, u=u)

    def qp_solve(self):
        xq = np.zeros((self.B, self.n_max))
        cvx = np.zeros(self.B, np.int32)
        rec = (abi.QpRecord * self.B)()
        self._chk(self.lib.tmx_qp_solve(self.h, _ptr(xq), _ptr(cvx), rec))
        return xq, cvx, rec

    def qp_solve_batched(self, qps, settings: abi.OsqpSettings = None):
        """sco::Model::optimize / trajopt_sqp::QPSolver::solve for a batch of QPs in CSC form (tmx_qp_solve_batched).
        qps: list of dicts with n, m, P_p, P_i, P_x, q, A_p, A_i, A_x, l, u and optionally x_warm, y_warm.
        Returns a list of dicts x, y, cvx_status, info (abi.QpInfo), active."""
        B = len(qps)
        arr = (abi.QpCsc * B)()
        keep = []

        def _a(v, dt):
            a = np.ascontiguousarray(v, dtype=dt)
            keep.append(a)
            return a

        for b, q in enumerate(qps):
            c = arr[b]
            c.n, c.m = int(q["n"]), int(q["m"])
            for k, dt, ct in (("P_p", np.int64, C.c_int64), ("P_i", np.int64, C.c_int64), ("P_x", np.float64, C.c_double),
                              ("q", np.float64, C.c_double), ("A_p", np.int64, C.c_int64), ("A_i", np.int64, C.c_int64),
                              ("A_x", np.float64, C.c_double), ("l", np.float64, C.c_double), ("u", np.float64, C.c_double)):
                setattr(c, k, _a(q[k], dt).ctypes.data_as(C.POINTER(ct)))
            if q.get("x_warm") is not None:
                c.x_warm = _a(q["x_warm"], np.float64).ctypes.data_as(C.POINTER(C.c_double))
                c.y_warm = _a(q["y_warm"], np.float64).ctypes.data_as(C.POINTER(C.c_double))
        ns, ms = [int(q["n"]) for q in qps], [int(q["m"]) for q in qps]
        x, y = np.zeros(max(1, sum(ns))), np.zeros(max(1, sum(ms)))
        act = np.zeros(max(1, sum(ms)), np.int32)
        cvx = np.zeros(B, np.int32)
        info = (abi.QpInfo * B)()
        self._chk(self.lib.tmx_qp_solve_batched(self.h, arr, B, C.byref(settings) if settings is not None else None, _ptr(x), _ptr(y),
                                                _ptr(cvx), info, _ptr(act)))
        out, on, om = [], 0, 0
        for b in range(B):
            out.append(dict(x=x[on:on + ns[b]].copy(), y=y[om:om + ms[b]].copy(), cvx_status=int(cvx[b]), info=info[b],
                            active=act[om:om + ms[b]].copy()))
            on += ns[b]
            om += ms[b]
        return out

    def qp_duals(self):
        """dual solution of the last qp_solve() / SQP step, reference row order: [B][m_max]"""
        y = np.zeros((self.B, self.m_max))
        self._chk(self.lib.tmx_qp_duals(self.h, _ptr(y)))
        return y

    def qp_active_set(self):
        """polish active-set flags of the last qp_solve() / SQP step, reference row order: [B][m_max] of -1 / 0 / +1"""
        f = np.zeros((self.B, self.m_max), np.int32)
        self._chk(self.lib.tmx_qp_active_set(self.h, _ptr(f)))
        return f

    def argmin(self, global_offset: int = 0):
        bi, bc = C.c_int64(), C.c_double()
        self._chk(self.lib.tmx_argmin(self.h, global_offset, C.byref(bi), C.byref(bc)))
        return bi.value, bc.value

    def best_trajectory(self):
        """the winning trajectory of the last argmin() on every rank (one broadcast from the owner rank): (x[T][D], owner rank)"""
        x = np.zeros((self.T, self.D))
        owner = C.c_int32(-1)
        self._chk(self.lib.tmx_best_trajectory(self.h, _ptr(x), C.byref(owner)))
        return x, owner.value

    def nccl_unique_id(self) -> bytes:
        buf = (C.c_uint8 * 128)()
        self._chk(self.lib.tmx_nccl_unique_id(buf))
        return bytes(buf)

    def nccl_init(self, unique_id: bytes, n_ranks: int, rank: int):
        """the library's own RCCL communicator for the best-seed reduction (tmx_argmin then spans all ranks)"""
        buf = (C.c_uint8 * 128)(*unique_id)
        self._chk(self.lib.tmx_nccl_init(self.h, buf, n_ranks, rank))

    def attach_nccl(self, comm_ptr: int):
        self._chk(self.lib.tmx_attach_nccl(self.h, C.c_void_p(comm_ptr)))

    def kernel_stats(self, reset: bool = False):
        a, n, c, e = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
        self._chk(self.lib.tmx_kernel_stats(self.h, C.byref(a), C.byref(n), C.byref(c), C.byref(e)))
        if reset:
            self._chk(self.lib.tmx_kernel_stats_reset(self.h))
        return dict(admm_ms=a.value, admm_launches=n.value, convexify_ms=c.value, evaluate_ms=e.value)


class BatchedTrustRegionSQP:
    """sco::BasicTrustRegionSQP for a batch of seeds (optimizers.hpp:137-194): setParameters, initialize, optimize,
    results.  `prob` is a trajopt_amd.problem.ProblemConstructionInfo (the hatched-problem description)."""

    def __init__(self, pci, device: int = 0, lib_path: str = None):
        self.pci = pci
        self.ctx = Context(device, lib_path)
        self.params = abi.default_sqp_params()
        self.osqp = abi.default_osqp_settings()
        self._uploaded = False
        self._callbacks = []
        self._step_callbacks = []

    def setParameters(self, params: abi.SqpParams):
        self.params = params
        self._uploaded = False

    def getParameters(self) -> abi.SqpParams:
        return self.params

    def initialize(self, x):
        """Optimizer::initialize (optimizers.cpp:127-136): x is [batch][n_steps][n_dof]; a wrong size raises."""
        x = np.asarray(x, dtype=np.float64)
        T, D = self.pci.basic_info.n_steps, self.pci.robot.n_dof + (1 if self.pci.basic_info.use_time else 0)
        if x.ndim != 3 or x.shape[1] != T or x.shape[2] != D:
            raise TmxError(f"initialization vector has wrong length. expected [B,{T},{D}] got {list(x.shape)}")
        if not self._uploaded:
            self.desc = self.pci.to_desc()
            self.ctx.upload(self.desc, self.params, self.osqp)
            self._uploaded = True
        self.ctx.set_x0(x)

    def addCallback(self, cb):
        """Optimizer::addCallback (optimizers.hpp:83-84).  cb(problem_index, results) with results = dict(x, status,
        total_cost, cost_vals, cnt_viols, n_func_evals, n_qp_solves, sqp_iter, merit_increases, trust_box_size): called
        before every SQP iteration of every seed and once more when the seed finishes, like the reference
        (optimizers.cpp:754, :978).  With callbacks the batch is stepped one trust-region evaluation per launch instead of
        running in the persistent kernel - an observability mode, not the fast path."""
        self._callbacks.append(cb)

    def addStepCallback(self, cb):
        """cb(problem_index, step) after EVERY trust-region evaluation (QP solve + exact re-evaluation) of every seed, with
        step = Context.step_log()[problem_index]: what BasicTrustRegionSQPResults::update leaves for ::print and the log writers
        (optimizers.cpp:380-647) - old / model / new values per cost and constraint, merit coefficients, the merits, dapprox,
        dexact, ratio.  Like addCallback it puts the optimizer into the stepped mode."""
        self._step_callbacks.append(cb)

    def _fire(self, b, r, cv, vv, st):
        res = dict(x=r["x"][b], status=int(r["status"][b]), total_cost=float(r["total_cost"][b]), cost_vals=cv[b], cnt_viols=vv[b],
                   n_func_evals=int(r["n_func_evals"][b]), n_qp_solves=int(r["n_qp_solves"][b]), sqp_iter=int(st["sqp_iter"][b]),
                   merit_increases=int(st["merit_increases"][b]), trust_box_size=float(st["trust_box_size"][b]))
        for cb in self._callbacks:
            cb(b, res)

    def optimize(self):
        if not self._callbacks and not self._step_callbacks:
            self.ctx.run(0)
            return self.ctx.results()["status"]
        B = self.ctx.B
        seen_iter = np.full(B, -1, np.int64)       # last (merit_increases, sqp_iter) whose start was reported
        finished = np.zeros(B, bool)
        n_qp_seen = np.zeros(B, np.int64)
        while True:
            r, st = self.ctx.results(), self.ctx.state()
            if self._step_callbacks:
                logs = self.ctx.step_log()
                for b in range(B):
                    if r["n_qp_solves"][b] > n_qp_seen[b]:
                        n_qp_seen[b] = r["n_qp_solves"][b]
                        for cb in self._step_callbacks:
                            cb(b, logs[b])
            cv, vv = self.ctx.evaluate()
            for b in range(B):
                if finished[b]:
                    continue
                if st["done"][b]:
                    finished[b] = True
                    self._fire(b, r, cv, vv, st)        # "at exit"
                elif st["merit_increases"][b] * 100000 + st["sqp_iter"][b] != seen_iter[b]:
                    seen_iter[b] = st["merit_increases"][b] * 100000 + st["sqp_iter"][b]
                    self._fire(b, r, cv, vv, st)        # before this SQP iteration starts
            if finished.all():
                break
            self.ctx.run(1)
        return self.ctx.results()["status"]

    def results(self):
        return self.ctx.results()


class BatchedTrustRegionSQPSolver(BatchedTrustRegionSQP):
    """trajopt_sqp::TrustRegionSQPSolver (trajopt_optimizers/trajopt_sqp/src/trust_region_sqp_solver.cpp:87-439) for a batch of seeds:
    the problem description must be of the trajopt_sqp flavour (pci.flavor = abi.FLAVOR_SQP).  registerCallback mirrors
    TrustRegionSQPSolver::registerCallback (:81): cb(problem_index, sqp_results) -> bool runs after every trust-region evaluation of
    every seed (stepSQPSolver, :421-422) with sqp_results = the fields of trajopt_sqp::SQPResults the step leaves (types.h:143-209):
    best_var_vals, best_costs / new_costs / new_approx_costs, best / new / new_approx constraint violations, merit_error_coeffs,
    box_size, best_exact_merit / new_approx_merit / new_exact_merit, approx / exact merit improvement and their ratio,
    penalty_iteration, convexify_iteration.  A callback that returns False ends THAT seed with SQPStatus::kStoppedByCallback
    (:432-436); the rest of the batch goes on.  With callbacks the batch is stepped one trust-region evaluation per launch."""

    def __init__(self, pci, device: int = 0, lib_path: str = None):
        if int(getattr(pci, "flavor", 0)) != abi.FLAVOR_SQP:
            raise TmxError("BatchedTrustRegionSQPSolver drives problems of the trajopt_sqp flavour (pci.flavor = abi.FLAVOR_SQP)")
        super().__init__(pci, device=device, lib_path=lib_path)
        self._sqp_callbacks = []

    def registerCallback(self, cb):
        self._sqp_callbacks.append(cb)

    def solve(self):
        """TrustRegionSQPSolver::solve (:87-159); returns the SQPStatus of every seed"""
        if not self._sqp_callbacks:
            self.ctx.run(0)
            return self.ctx.results()["status"]
        B = self.ctx.B
        n_qp_seen = np.zeros(B, np.int64)
        stopped = np.zeros(B, bool)
        while True:
            self.ctx.run(1)
            r, st, logs = self.ctx.results(), self.ctx.state(), self.ctx.step_log()
            for b in range(B):
                if stopped[b] or r["n_qp_solves"][b] <= n_qp_seen[b] or not logs[b]["valid"]:
                    continue
                n_qp_seen[b] = r["n_qp_solves"][b]
                lg = logs[b]
                res = dict(best_var_vals=r["x"][b].reshape(-1), best_costs=lg["old_cost_vals"], new_costs=lg["new_cost_vals"],
                           new_approx_costs=lg["model_cost_vals"], best_constraint_violations=lg["old_cnt_viols"],
                           new_constraint_violations=lg["new_cnt_viols"], new_approx_constraint_violations=lg["model_cnt_viols"],
                           merit_error_coeffs=lg["merit_error_coeffs"], box_size=lg["box_size"], best_exact_merit=lg["old_merit"],
                           new_approx_merit=lg["model_merit"], new_exact_merit=lg["new_merit"], approx_merit_improve=lg["approx_merit_improve"],
                           exact_merit_improve=lg["exact_merit_improve"], merit_improve_ratio=lg["merit_improve_ratio"],
                           penalty_iteration=lg["merit_increases"], convexify_iteration=lg["sqp_iter"], n_qp_solves=int(r["n_qp_solves"][b]))
                ok = True
                for cb in self._sqp_callbacks:
                    ok = bool(cb(b, res)) and ok       # (every callback runs: success &= callback->execute(...), :444-445)
                if not ok and not st["done"][b]:
                    self.ctx.stop(b, abi.SQP_STOPPED_BY_CALLBACK)
                    stopped[b] = True
            if (st["done"] | stopped).all():
                break
        return self.ctx.results()["status"]


def OptimizeProblem(pci, init_traj, device: int = 0, lib_path: str = None):
    """trajopt::OptimizeProblem (problem_description.cpp:396-408): BasicTrustRegionSQP with the reference's planner-style
    parameters on the given initial trajectory [n_steps][n_dof]; returns the fields of trajopt::TrajOptResult"""
    opt = BatchedTrustRegionSQP(pci, device=device, lib_path=lib_path)
    p = opt.getParameters()
    p.max_iter, p.min_approx_improve_frac, p.improve_ratio_threshold, p.initial_merit_error_coeff = 40, 0.001, 0.2, 20.0
    opt.initialize(np.asarray(init_traj, dtype=np.float64)[None, :, :])
    opt.optimize()
    r = opt.results()
    cv, vv = opt.ctx.evaluate()
    opt.ctx.close()
    return dict(cost_names=pci.cost_names(), cnt_names=pci.cnt_names(), cost_vals=cv[0], cnt_viols=vv[0], traj=r["x"][0],
                status=int(r["status"][0]))
