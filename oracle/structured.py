"""NumPy front end of oracle/mgcmt_oracle.c (matrix-free CPU restatement).

TEST INFRASTRUCTURE — NOT PRODUCT CODE (see oracle/__init__.py).  Used for parity checks at sizes the
generic-sparse oracle is too slow for, and as the timed CPU baseline of bench.py ("port").
"""
import ctypes
import os
import subprocess
import time
from ctypes import POINTER, c_double, c_int, c_long

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmgcmt_oracle.so")
_dp = POINTER(c_double)
_lib = None

WJACOBI, GS_LEX, SOR_LEX, GS_MC = 0, 1, 2, 3


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "mgcmt_oracle.c")
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
        L = ctypes.CDLL(_SO)
        L.mgo_threads.restype = c_int
        L.mgo_galerkin.argtypes = [_dp, c_long, _dp]
        L.mgo_vcycle.restype = c_int
        L.mgo_vcycle.argtypes = [c_int, c_long, c_long, c_int, _dp, _dp, c_double, c_int, c_double, c_int, c_int, c_int, _dp, _dp]
        L.mgo_apply_level.argtypes = [c_int, c_long, c_long, c_int, _dp, _dp, c_double, _dp, _dp]
        L.mgo_smooth_level.argtypes = [c_int, c_long, c_long, c_int, _dp, _dp, c_double, c_int, c_double, c_int, _dp, _dp]
        L.mgo_residual_level.argtypes = [c_int, c_long, c_long, c_int, _dp, _dp, c_double, _dp, _dp]
        L.mgo_restrict.argtypes = [c_int, c_long, c_long, _dp, _dp]
        L.mgo_prolong.argtypes = [c_int, c_long, c_long, _dp, _dp, c_int]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def tri_laplacian(n):
    """MGCMTStencilMaker.py:17-21 as a [3][n] factor."""
    h = 1. / n
    s = 1 / h ** 2
    t = np.zeros((3, n))
    t[0, 1:] = s
    t[1] = -2.0 * s
    t[2, :-1] = s
    return t


def tri_identity(n):
    t = np.zeros((3, n))
    t[1] = 1.0
    return t


def laplacian_factors(g, dimension, scale=1.0):
    """(X, Y) factor blocks [nterms][3][g] of scale * laplacian(g, dimension)."""
    L, I = tri_laplacian(g) * scale, tri_identity(g)
    if dimension == "1d":
        return None, np.ascontiguousarray(L[None])
    return np.ascontiguousarray(np.stack([I, L])), np.ascontiguousarray(np.stack([L, I]))


def galerkin(t):
    t = np.ascontiguousarray(t, dtype=np.float64)
    out = np.zeros((3, t.shape[1] // 2))
    lib().mgo_galerkin(_p(t), t.shape[1], _p(out))
    return out


def _dims(X, Y):
    nterms, _, nc = Y.shape
    return (1 if X is None else 2), (1 if X is None else X.shape[2]), nc, nterms


def apply(X, Y, shift, v):
    dim, nr, nc, nt = _dims(X, Y)
    v = np.ascontiguousarray(v, dtype=np.float64)
    out = np.zeros(nr * nc)
    lib().mgo_apply_level(dim, nr, nc, nt, _p(X), _p(Y), shift, _p(v), _p(out))
    return out


def smooth(X, Y, shift, kind, v, f, nu=4, omega=1.0):
    dim, nr, nc, nt = _dims(X, Y)
    v = np.array(v, dtype=np.float64).reshape(-1).copy()
    f = np.ascontiguousarray(f, dtype=np.float64).reshape(-1)
    lib().mgo_smooth_level(dim, nr, nc, nt, _p(X), _p(Y), shift, kind, omega, nu, _p(v), _p(f))
    return v


def residual(X, Y, shift, v, f):
    dim, nr, nc, nt = _dims(X, Y)
    v = np.ascontiguousarray(v, dtype=np.float64).reshape(-1)
    f = np.ascontiguousarray(f, dtype=np.float64).reshape(-1)
    r = np.zeros(nr * nc)
    lib().mgo_residual_level(dim, nr, nc, nt, _p(X), _p(Y), shift, _p(v), _p(f), _p(r))
    return r


def restrict(dim, nr, nc, fine):
    fine = np.ascontiguousarray(fine, dtype=np.float64).reshape(-1)
    out = np.zeros((nr // 2 if dim == 2 else 1) * (nc // 2))
    lib().mgo_restrict(dim, nr, nc, _p(fine), _p(out))
    return out


def prolong(dim, nr, nc, coarse):
    coarse = np.ascontiguousarray(coarse, dtype=np.float64).reshape(-1)
    out = np.zeros(nr * nc)
    lib().mgo_prolong(dim, nr, nc, _p(coarse), _p(out), 0)
    return out


def vcycle(X, Y, g, lowest, shift, kind, v0, f, nu1=4, nu2=4, nu_coarse=4, omega=1.0, inplace=False):
    dim = 1 if X is None else 2
    v = v0 if inplace else np.array(v0, dtype=np.float64).reshape(-1).copy()
    f = np.ascontiguousarray(f, dtype=np.float64).reshape(-1)
    rc = lib().mgo_vcycle(dim, g, lowest, Y.shape[0], _p(X), _p(Y), shift, kind, omega, nu1, nu2, nu_coarse, _p(v), _p(f))
    if rc != 0:
        raise RuntimeError("mgo_vcycle failed")
    return v


def time_cpu_baseline(smoother, nu, lowest, budget_seconds, grid=8192, workload_grid=None):
    """bench.py's cpu_baseline: whole V(nu,nu) cycles of the same workload on the host cores (OpenMP over
    all of them for the order-independent sweeps), for about budget_seconds."""
    kind, omega = (WJACOBI, 2. / 3.) if smoother == "wjacobi" else (GS_MC, 1.0)
    # a one-GPU box's CPU share is 16 cores (more threads than that only thrash one NUMA node)
    lib().mgo_set_threads(min(os.cpu_count() or 1, 16))
    threads = lib().mgo_threads()
    g = grid
    X, Y = laplacian_factors(g, "2d", scale=-1.0 / np.pi ** 2)
    rng = np.random.RandomState(1)
    f = rng.rand(g * g)
    v = np.zeros(g * g)
    t0 = time.perf_counter()
    vcycle(X, Y, g, lowest, 0.0, kind, v, f, nu, nu, nu, omega, inplace=True)     # first cycle also pages memory in
    first = time.perf_counter() - t0
    cycles, spent = 0, 0.0
    while spent < max(budget_seconds - first, 0.0) or cycles == 0:
        t0 = time.perf_counter()
        vcycle(X, Y, g, lowest, 0.0, kind, v, f, nu, nu, nu, omega, inplace=True)
        spent += time.perf_counter() - t0
        cycles += 1
    mlups = float(g) * g * 2 * nu * cycles / spent / 1e6
    return {"value": mlups, "unit": "MLUPS", "cores": threads, "kind": "port",
            "sample": "%d whole V(%d,%d) %s cycles on a %d^2 grid (%s) after "
                      "1 warm-up cycle, C restatement oracle/mgcmt_oracle.c, OpenMP x%d" %
                      (cycles, nu, nu, smoother, g,
                       "the full workload" if (workload_grid or g) == g else
                       "bounded sample of the %s^2 workload; MLUPS is per point" % workload_grid, threads),
            "vcycles_per_s": cycles / spent}
