"""The fused passes of 1-D levels (csrc/kernels_fused1d.hip: a wave takes a window of 128 points through every stage of
a pass in registers) against the one-launch-per-operation kernels (MGCMT_OPT_FUSED = 0) and against the C oracle
(MGCMTSolver.py:281-329 restated in oracle/mgcmt_oracle.c): constant operators with their Galerkin levels, a potential
on the diagonal (1DPotMatrixVcycle.py:16: the general tridiagonal path), shifts, several columns with Gram-Schmidt,
every sweep count a pass takes, recompute-instead-of-store, zero start."""
import numpy as np
import pytest

from conftest import rel_err
from multigridcmt_amd import _lib
from multigridcmt_amd.operators import StructuredOperator, laplacian_operator, tri_laplacian
from multigridcmt_amd.plan import Plan
from oracle import structured as st

SCALE = -1 / np.pi ** 2


def _op(n, potential):
    if not potential:
        return laplacian_operator(n, "1d") * SCALE, st.laplacian_factors(n, "1d", SCALE)[1]
    t = tri_laplacian(n) * SCALE
    t[1] += 3.0 * np.random.RandomState(21).rand(n)          # a potential well's diagonal
    return StructuredOperator("1d", n, [(None, t)]), np.ascontiguousarray(t[None])


def _cycle(n, fused, kind, omega, nu1, nu2, nuc, potential=False, k=1, gs=False, recompute=1, zero=False, lowest=8, level=0):
    op, _ = _op(n, potential)
    p = Plan(op, lowest, nvec=k)
    p.set_option(_lib.OPT_FUSED, fused)
    p.set_option(_lib.OPT_RECOMPUTE, recompute)
    p.set_shifts(0.3 + 0.1 * np.arange(k))
    rng = np.random.RandomState(3)
    nl = n >> level
    for q in range(k):
        v0 = np.zeros(nl) if zero else rng.rand(nl)
        p.upload(level, _lib.SLOT_V, q, v0 + (7.0 if zero else 0.0) * 0)
        p.upload(level, _lib.SLOT_F, q, rng.rand(nl))
    if zero:                                              # V holds garbage: the zero start is a flag
        for q in range(k):
            p.upload(level, _lib.SLOT_V, q, rng.rand(nl) * 1e3)
    p.vcycle(nu1, nu2, kind, omega=omega, k=k, nu_coarse=nuc, gram_schmidt=gs, level=level, zero_start=zero)
    out = np.stack([p.download(level, _lib.SLOT_V, q) for q in range(k)])
    kinds = [p.fused_max_sweeps(l, kind) for l in range(p.num_levels)]
    p.close()
    return out, kinds


@pytest.mark.parametrize("n", [256, 1024, 4096])
@pytest.mark.parametrize("kind,okind,omega", [(_lib.WJACOBI, st.WJACOBI, 2. / 3.), (_lib.GS_MC, st.GS_MC, 1.0), (_lib.GS_MC, st.GS_MC, 1.2)])
@pytest.mark.parametrize("potential", [False, True])
def test_fused_1d_cycle_against_oracle_and_unfused(backend, n, kind, okind, omega, potential):
    _, Y = _op(n, potential)
    rng = np.random.RandomState(3)
    v0, f = rng.rand(n), rng.rand(n)
    # two correct evaluation orders of one cycle differ by rounding times the condition of the coarse problems (~ n^2
    # here): below 1e-12 at n = 256, a few 1e-12 at n = 1024 and 4096 (north star: 1e-10)
    tol = {256: 1e-12, 1024: 5e-12, 4096: 5e-11}[n]
    for nu1, nu2, nuc in ((2, 2, 2), (4, 4, 4), (1, 3, 4), (5, 2, 3)):
        a, kinds = _cycle(n, 1, kind, omega, nu1, nu2, nuc, potential)
        assert kinds[0] >= 4 and kinds[-1] == 0 and kinds[-2] >= 4   # every level of >= 16 points is on the fused passes, the 8-point level is not
        b, _ = _cycle(n, 0, kind, omega, nu1, nu2, nuc, potential)
        assert rel_err(a, b) < tol, (nu1, nu2, nuc)
        y = st.vcycle(None, Y, n, 8, 0.3, okind, v0, f, nu1, nu2, nuc, omega)
        assert rel_err(a[0], y) < tol, (nu1, nu2, nuc)


@pytest.mark.parametrize("kind,omega", [(_lib.WJACOBI, 2. / 3.), (_lib.GS_MC, 1.0)])
def test_fused_1d_recompute_zero_start_and_columns(backend, kind, omega):
    n = 2048
    ref, _ = _cycle(n, 0, kind, omega, 2, 2, 2, k=3, gs=True)
    for recompute in (1, 2):                               # 2: the no-store down pass + recomputing up pass on every level
        a, _ = _cycle(n, 1, kind, omega, 2, 2, 2, k=3, gs=True, recompute=recompute)
        assert rel_err(a, ref) < 1e-10, recompute
        assert np.allclose(a @ a.T, np.eye(3), atol=1e-12)
    for recompute in (1, 2):
        z0, _ = _cycle(n, 0, kind, omega, 2, 2, 2, zero=True, recompute=recompute)
        z1, _ = _cycle(n, 1, kind, omega, 2, 2, 2, zero=True, recompute=recompute)
        assert rel_err(z1, z0) < 1e-12, recompute
    # a sub-cycle from a Galerkin level (constant but for its last diagonal entry)
    s1, _ = _cycle(n, 1, kind, omega, 3, 3, 2, level=2, recompute=2)
    s0, _ = _cycle(n, 0, kind, omega, 3, 3, 2, level=2)
    assert rel_err(s1, s0) < 1e-12


@pytest.mark.parametrize("kind,okind,omega", [(_lib.WJACOBI, st.WJACOBI, 2. / 3.), (_lib.GS_MC, st.GS_MC, 1.0)])
def test_fused_1d_smoothing_and_transfers_alone(backend, kind, okind, omega):
    """mgcmt_smooth on 1-D levels (every sweep count a pass takes and more) and the pass modes one by one, on the finest
    level and on a Galerkin level, against the oracle's level functions."""
    n = 1024
    op, Yf = _op(n, False)
    rng = np.random.RandomState(9)
    for level in (0, 1):
        nl = n >> level
        Y = Yf if level == 0 else np.ascontiguousarray(st.galerkin(Yf[0])[None])
        v0, f = rng.rand(nl), rng.rand(nl)
        p = Plan(op, 8, nvec=1)
        p.set_shifts([0.45])
        for nu in (1, 2, 4, 7, 13):
            p.upload(level, _lib.SLOT_V, 0, v0)
            p.upload(level, _lib.SLOT_F, 0, f)
            p.smooth(level, kind, nu, omega)
            want = st.smooth(None, Y, 0.45, okind, v0, f, nu, omega)
            assert rel_err(p.download(level, _lib.SLOT_V, 0), want) < 1e-12, (level, nu)
        # residual + restriction fused behind two sweeps (mode 2), interpolation + correction in front of them (mode 1)
        p.upload(level, _lib.SLOT_V, 0, v0)
        p.upload(level, _lib.SLOT_F, 0, f)
        p.fused_pass(level, kind, 2, omega, mode=2)
        v2 = st.smooth(None, Y, 0.45, okind, v0, f, 2, omega)
        assert rel_err(p.download(level, _lib.SLOT_V, 0), v2) < 1e-12
        rc = st.restrict(1, 1, nl, st.residual(None, Y, 0.45, v2, f))
        assert rel_err(p.download(level + 1, _lib.SLOT_F, 0), rc) < 1e-12
        e = rng.rand(nl // 2)
        p.upload(level + 1, _lib.SLOT_V, 0, e)
        p.upload(level, _lib.SLOT_V, 0, v0)
        p.fused_pass(level, kind, 2, omega, mode=1)
        want = st.smooth(None, Y, 0.45, okind, v0 + st.prolong(1, 1, nl, e), f, 2, omega)
        assert rel_err(p.download(level, _lib.SLOT_V, 0), want) < 1e-12
        p.close()
