"""include/tmx_detmath.h — the sin / cos / atan2 shared by the device kernels and the oracle (fixed IEEE operation
sequence).  CPU tier: accuracy against numpy (glibc) and bit-identity of the oracle's and the host-built kernels' copies;
GPU tier: the device build returns the SAME BITS as the oracle's build on 1e6 points (that is what makes term values,
FD Jacobians and therefore nnz(A) identical across host and device)."""
import ctypes as C

import numpy as np
import pytest


def _inputs(n=200_000, seed=0):
    rng = np.random.default_rng(seed)
    a = np.concatenate([rng.uniform(-4, 4, n), rng.uniform(-50, 50, n), rng.normal(0, 1e-3, n), rng.uniform(-1e5, 1e5, n),
                        rng.integers(-400, 400, n) * (np.pi / 2) + rng.normal(0, 1e-9, n)])
    b = np.concatenate([rng.normal(0, 1, 3 * n), np.abs(rng.normal(0, 1e-3, n)), rng.uniform(0, 1, n)])
    return np.ascontiguousarray(a), np.ascontiguousarray(b)


def _orc(orc, op, a, b):
    out = np.empty_like(a)
    P = C.POINTER(C.c_double)
    assert orc.lib().orc_detmath(op, len(a), a.ctypes.data_as(P), b.ctypes.data_as(P), out.ctypes.data_as(P)) == 0
    return out


def _dev(ctx, op, a, b):
    out = np.empty_like(a)
    fn = ctx.lib.tmx_debug_detmath
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    fn.restype = C.c_int
    assert fn(ctx.h, op, len(a), a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)) == 0
    return out


def _ulps(got, ref):
    return np.abs(got - ref) / np.spacing(np.abs(ref))


def test_accuracy_against_libm(orc):
    a, b = _inputs()
    assert _ulps(_orc(orc, 0, a, b), np.sin(a)).max() <= 1.0
    assert _ulps(_orc(orc, 1, a, b), np.cos(a)).max() <= 1.0
    assert _ulps(_orc(orc, 2, a, b), np.arctan2(a, b)).max() <= 2.0
    # special values follow C99 Annex F
    ys = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, 0.0, 0.0, -0.0, 3.0, -3.0])
    xs = np.array([1.0, 1.0, 0.0, 0.0, np.inf, -np.inf, -1.0, 0.0, -0.0, -np.inf, np.inf])
    got, ref = _orc(orc, 2, ys, xs), np.arctan2(ys, xs)
    assert np.array_equal(got, ref) and np.array_equal(np.signbit(got), np.signbit(ref))
    assert np.isnan(_orc(orc, 0, np.array([np.inf, np.nan]), np.zeros(2))).all()


def test_hostemu_copy_is_bit_identical(orc, hostemu_lib):
    from trajopt_amd import runtime
    ctx = runtime.Context(0, hostemu_lib)
    a, b = _inputs(20_000, seed=1)
    for op in (0, 1, 2):
        assert np.array_equal(_dev(ctx, op, a, b).view(np.uint64), _orc(orc, op, a, b).view(np.uint64))
    ctx.close()


@pytest.mark.gpu
def test_device_copy_is_bit_identical(orc, gpu_ctx_factory):
    ctx = gpu_ctx_factory()
    a, b = _inputs(200_000, seed=2)
    for op in (0, 1, 2):
        assert np.array_equal(_dev(ctx, op, a, b).view(np.uint64), _orc(orc, op, a, b).view(np.uint64)), f"op {op}"
    ctx.close()
