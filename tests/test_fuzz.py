"""Randomised A/B of the optimised cycle (fused passes, recompute, in-place colour windows, LDS tail, graph replay)
against the one-launch-per-operation kernels over grid sizes, coarsest levels, sweep counts, smoothers, relaxation
factors, shifts, vector counts and option settings.  A few cases on the emulator, many on the GPU."""
import os

import numpy as np
import pytest

from conftest import rel_err
from multigridcmt_amd import _lib
from multigridcmt_amd.operators import laplacian_operator, potential_well_operator
from multigridcmt_amd.plan import Plan


def _case(rng, sizes, dense_tail=False):
    g = int(rng.choice(sizes))
    lowest = int(rng.choice([s for s in (2, 4, 8, 16) if s < g]))
    kind = int(rng.choice([_lib.WJACOBI, _lib.GS_MC]))
    omega = float(rng.choice([2. / 3., 0.8])) if kind == _lib.WJACOBI else float(rng.choice([1.0, 1.15]))
    nu1, nu2, nuc = (int(x) for x in rng.randint(0, 4, size=3))
    k = int(rng.randint(1, 4))
    well = bool(rng.randint(0, 3) == 0)
    opts = {_lib.OPT_RECOMPUTE: int(rng.choice([0, 1, 2])), _lib.OPT_TAIL: int(rng.randint(0, 3) if dense_tail else rng.choice([0, 2])), _lib.OPT_GRAPH: int(rng.randint(0, 2))}
    zero_start = bool(rng.randint(0, 3) == 0)    # every cycle starts from "V is zero" as a flag (MGCMT_CYCLE_ZERO_START); V holds garbage
    return g, lowest, kind, omega, nu1, nu2, nuc, k, well, opts, zero_start


def _run_case(case, seed):
    g, lowest, kind, omega, nu1, nu2, nuc, k, well, opts, zero_start = case
    op = potential_well_operator(g, 20.0, (g // 4, 3 * g // 4)) if well else laplacian_operator(g, "2d") * (-1 / np.pi ** 2)
    rng = np.random.RandomState(seed)
    v0, f = rng.rand(k, g * g), rng.rand(k, g * g)
    outs = []
    for fused in (1, 0):
        p = Plan(op, lowest, nvec=k)
        p.set_option(_lib.OPT_FUSED, fused)
        for o, val in opts.items():
            p.set_option(o, val)
        p.set_shifts(0.2 + 0.5 * np.arange(k))
        for q in range(k):
            p.upload(0, _lib.SLOT_V, q, v0[q] * (1e6 if zero_start else 1.0))
            p.upload(0, _lib.SLOT_F, q, f[q])
        for _ in range(3):                      # with graph replay from the second call on
            p.vcycle(nu1, nu2, kind, omega=omega, k=k, nu_coarse=nuc, zero_start=zero_start)
        outs.append(np.stack([p.download(0, _lib.SLOT_V, q) for q in range(k)]))
        p.close()
    return rel_err(outs[0], outs[1])


def test_random_cycles_small(backend):
    rng = np.random.RandomState(int(os.environ.get("MGCMT_FUZZ_SEED", 2024 if backend == "emu" else 7)))
    n, sizes = (6, (32, 64, 128)) if backend == "emu" else (60, (32, 64, 128, 256, 512, 1024, 2048))
    n = int(os.environ.get("MGCMT_FUZZ_CASES", n))          # longer soak runs: MGCMT_FUZZ_CASES=500 MGCMT_FUZZ_SEED=...
    for i in range(n):
        case = _case(rng, sizes, dense_tail=backend == "hip")
        err = _run_case(case, 100 + i)
        assert err < 1e-11, (case, err)


def _rq_case(rng, sizes_1d, sizes_2d):
    dim = "1d" if rng.randint(0, 2) == 0 else "2d"
    g = int(rng.choice(sizes_1d if dim == "1d" else sizes_2d))
    kind = int(rng.randint(0, 3))                     # 0: scaled Laplacian, 1: + potential (a well), 2: the same with a mass operator
    nu = int(rng.randint(0, 6))
    cycle = bool(rng.randint(0, 2))                   # vcycle_rqmg (Galerkin pairs on every level) or rqmin alone
    nmin = int(rng.choice([s for s in (2, 4, 8) if s < g] or [2]))
    return dim, g, kind, nu, cycle, nmin


def _rq_operators(dim, g, kind):
    import scipy.sparse as sp
    from multigridcmt_amd import MGCMTStencilMaker
    sm = MGCMTStencilMaker()
    A = ((-1 / np.pi ** 2) * sm.laplacian(g, dimension=dim)).tocsr()
    m1 = sp.diags([np.full(g - 1, 1 / 6), np.full(g, 2 / 3), np.full(g - 1, 1 / 6)], [-1, 0, 1])
    if kind >= 1:
        chi = np.zeros(g)
        chi[g // 4:3 * g // 4] = 1.0
        V = 20.0 * (1.0 - chi) if dim == "1d" else 20.0 * (1.0 - np.outer(chi, chi)).reshape(-1)
        A = (A + sp.diags(V)).tocsr()
    n = g if dim == "1d" else g * g
    M = sp.eye(n, format="csr") if kind < 2 else (m1 if dim == "1d" else sp.kron(m1, m1)).tocsr()
    return A, M


def test_random_rayleigh_quotient_minimisations(backend):
    """rqmin / vcycle_rqmg on the device (csrc/kernels_rq.hip: the single-launch form of small levels, the one-thread-per-
    point passes of 1-D and odd levels, the row marches with their 5-point and general-term forms, <g, M g> by a march or by
    application + dot product) against the CPU restatement of MGCMTSolver.py:17-57,99-122 over random sizes, operators,
    mass operators, step counts and coarsest levels.  Tolerances: the steps amplify rounding differences."""
    from oracle.sparse_ref import RefSolver
    from multigridcmt_amd import MGCMTSolver
    small = backend == "emu" and os.environ.get("MGCMT_FUZZ_ALL_SIZES", "0") != "1"   # (the GPU's case list on the emulator: slow)
    n_cases, s1, s2 = (5, (8, 64, 256), (4, 16, 32)) if small else (40, (4, 8, 64, 512, 2048, 8192), (8, 16, 32, 64, 128))
    rng = np.random.RandomState(int(os.environ.get("MGCMT_FUZZ_SEED", 11 if small else 12)))
    n_cases = int(os.environ.get("MGCMT_FUZZ_CASES", n_cases))
    solver, ref = MGCMTSolver(), RefSolver()
    for i in range(n_cases):
        case = _rq_case(rng, s1, s2)
        dim, g, kind, nu, cycle, nmin = case
        A, M = _rq_operators(dim, g, kind)
        x0 = np.random.RandomState(200 + i).rand(A.shape[0])
        import warnings
        use_cycle = cycle and g > nmin
        if use_cycle:
            nu = max(nu, 1)
            x, rho = solver.vcycle_rqmg(x0.copy(), A, M, nu1=nu, nu2=nu, nmin=nmin)
        else:
            x, rho = solver.rqmin(A, x0.copy(), M, nu=nu)
        assert np.all(np.isfinite(x)) and np.isfinite(rho), case
        # two comparisons: tightly with the CPU restatement whose 2 x 2 pencils are solved in closed form (extended
        # precision), loosely with the faithful one (LAPACK's QZ, as the reference): QZ's delta carries a forward error of up
        # to ~1e-9 on the badly scaled pencils of a random start vector on a fine grid (oracle/sparse_ref.py: exact_pencil)
        for exact, tol_rho, tol_x in ((True, 1e-10, 1e-8), (False, 1e-6, 1e-5)):
            ref.exact_pencil = exact
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                try:
                    if use_cycle:
                        xr, rr = ref.vcycle_rqmg(x0.copy(), A, M, nu1=nu, nu2=nu, nmin=nmin, dimension=dim)
                    else:
                        xr, rr = ref.rqmin(A, x0.copy(), M, nu=nu)
                except ValueError:                   # scipy's eig refusing the NaNs of a degenerate pencil (below)
                    xr, rr = np.full_like(x0, np.nan), np.nan
                finally:
                    ref.exact_pencil = False
            xr, rr = np.real(xr), float(np.real(rr))
            if not (np.all(np.isfinite(xr)) and np.isfinite(rr)):
                # the reference's arithmetic has no answer here: a gradient of exact zeros (x an eigenvector to the last bit, as
                # on a 2-point level after one step) makes its 2 x 2 pencil singular and eig returns NaNs; the device code
                # leaves x as it is — nothing to compare with, but the result must be a Rayleigh quotient of x
                assert abs(rho - x @ (A @ x) / (x @ (M @ x))) < 1e-9 * abs(rho), case
                continue
            assert abs(rho - rr) < tol_rho * abs(rr), (case, exact, rho, rr)
            assert rel_err(x, xr) < tol_x, (case, exact, rel_err(x, xr))
