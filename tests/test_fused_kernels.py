"""The fused row-streaming kernels (csrc/kernels_fused.hip) against the one-launch-per-operation
kernels (A/B through MGCMT_OPT_FUSED) and against the C oracle.  Both backends."""
import numpy as np
import pytest

from conftest import rel_err
from multigridcmt_amd import _lib
from multigridcmt_amd.operators import laplacian_operator
from multigridcmt_amd.plan import Plan
from oracle import structured as st

SCALE = -1 / np.pi ** 2


def _run(g, fused, kind, omega, what, nu, shift=0.7, k=1, gs=False, seed=0, rows=0, recompute=1):
    op = laplacian_operator(g, "2d") * SCALE
    rng = np.random.RandomState(seed)
    p = Plan(op, 8, nvec=k)
    p.set_option(_lib.OPT_FUSED, fused)
    p.set_option(_lib.OPT_FUSED_ROWS, rows)      # rows a wave marches over (0 = automatic); process-wide
    p.set_option(_lib.OPT_RECOMPUTE, recompute)  # 2: recompute-instead-of-store on every fused level
    p.set_shifts(np.full(k, shift) + 0.1 * np.arange(k))
    for q in range(k):
        p.upload(0, _lib.SLOT_V, q, rng.rand(g * g))
        p.upload(0, _lib.SLOT_F, q, rng.rand(g * g))
    if what == "smooth":
        p.smooth(0, kind, nu, omega, k=k)
    else:
        p.vcycle(nu, nu, kind, omega=omega, k=k, nu_coarse=nu, gram_schmidt=gs)
    out = np.stack([p.download(0, _lib.SLOT_V, q) for q in range(k)])
    coarse = p.download(1, _lib.SLOT_F, 0) if what == "vcycle" else None
    p.close()
    return out, coarse


@pytest.mark.parametrize("g", [128, 256])
@pytest.mark.parametrize("kind,omega", [(_lib.WJACOBI, 2. / 3.), (_lib.GS_MC, 1.0), (_lib.GS_MC, 1.3)])
def test_fused_equals_unfused(backend, g, kind, omega):
    for what in ("smooth", "vcycle"):
        for nu in (1, 2, 3):
            a, ca = _run(g, 1, kind, omega, what, nu)
            b, cb = _run(g, 0, kind, omega, what, nu)
            assert rel_err(a, b) < 1e-12, (what, nu)
            if ca is not None:
                assert rel_err(ca, cb) < 1e-12, (what, nu)        # the fused residual + restriction


@pytest.mark.parametrize("kind,omega", [(_lib.WJACOBI, 2. / 3.), (_lib.GS_MC, 1.1)])
def test_fused_chunk_lengths(backend, kind, omega):
    """The marching loop has a checked body (chunk ends, grid boundary) and an unchecked steady-state one; short
    chunks run only the first, long ones mostly the second.  Every chunk length gives the same cycle."""
    ref, cref = _run(256, 0, kind, omega, "vcycle", 2)
    try:
        for rows in (6, 22, 64, 256):
            for recompute in (1, 2):
                a, ca = _run(256, 1, kind, omega, "vcycle", 2, rows=rows, recompute=recompute)
                assert rel_err(a, ref) < 1e-12, (rows, recompute)
                assert rel_err(ca, cref) < 1e-12, (rows, recompute)
    finally:
        _run(16, 1, kind, omega, "smooth", 1, rows=0)     # back to automatic


def test_fused_multi_vector_cycle(backend):
    a, _ = _run(128, 1, _lib.WJACOBI, 2. / 3., "vcycle", 2, k=3, gs=True)
    b, _ = _run(128, 0, _lib.WJACOBI, 2. / 3., "vcycle", 2, k=3, gs=True)
    assert rel_err(a, b) < 1e-10
    assert np.allclose(a @ a.T, np.eye(3), atol=1e-12)


@pytest.mark.parametrize("kind,okind,omega", [(_lib.WJACOBI, st.WJACOBI, 2. / 3.), (_lib.GS_MC, st.GS_MC, 1.0)])
def test_fused_cycle_against_c_oracle(backend, kind, okind, omega):
    g = 256
    rng = np.random.RandomState(0)
    v0, f = rng.rand(g * g), rng.rand(g * g)       # same draws as _run(seed=0)
    a, _ = _run(g, 1, kind, omega, "vcycle", 2)
    X, Y = st.laplacian_factors(g, "2d", SCALE)
    y = st.vcycle(X, Y, g, 8, 0.7, okind, v0, f, 2, 2, 2, omega)
    assert rel_err(a[0], y) < 1e-10


@pytest.mark.parametrize("kind,omega", [(_lib.WJACOBI, 2. / 3.), (_lib.GS_MC, 1.0), (_lib.GS_MC, 1.2)])
def test_fused_nine_point_level(backend, kind, omega):
    """Direct A/B on a Galerkin (separable 9-point) level: smoothing, residual+restriction, prolongation."""
    g = 512
    op = laplacian_operator(g, "2d") * SCALE
    rng = np.random.RandomState(7)
    v1, f1 = rng.rand(256 * 256), rng.rand(256 * 256)
    outs = []
    for fused in (1, 0):
        p = Plan(op, 8, nvec=1)
        p.set_option(_lib.OPT_FUSED, fused)
        p.set_shifts([0.9])
        res = []
        for nu in (1, 2, 3):
            p.upload(1, _lib.SLOT_V, 0, v1)
            p.upload(1, _lib.SLOT_F, 0, f1)
            p.smooth(1, kind, nu, omega)
            res.append(p.download(1, _lib.SLOT_V, 0))
        p.upload(1, _lib.SLOT_V, 0, v1)
        p.upload(1, _lib.SLOT_F, 0, f1)
        p.vcycle(2, 1, kind, omega=omega, nu_coarse=3, level=1)       # sub-cycle starting on the 9-point level
        res.append(p.download(1, _lib.SLOT_V, 0))
        res.append(p.download(2, _lib.SLOT_F, 0))
        outs.append(res)
        p.close()
    for a, b in zip(*outs):
        assert rel_err(a, b) < 1e-12


@pytest.mark.parametrize("kind,omega", [(_lib.WJACOBI, 2. / 3.), (_lib.GS_MC, 1.0)])
def test_fused_general_separable_operator(backend, kind, omega):
    """Laplacian plus a separable potential V(x,y) = a(x) + b(y): variable tridiagonal factors, so every
    level runs the general separable 9-point policy (Op9<2>) instead of the constant-coefficient ones."""
    from multigridcmt_amd.operators import StructuredOperator, tri_identity, tri_laplacian
    g = 256
    rng = np.random.RandomState(11)
    Lx, Ly = tri_laplacian(g) * SCALE, tri_laplacian(g) * SCALE
    Lx[1] += 5.0 * rng.rand(g)
    Ly[1] += 5.0 * rng.rand(g)
    op = StructuredOperator("2d", g, [(tri_identity(g), Ly), (Lx, tri_identity(g))])
    v0, f = rng.rand(g * g), rng.rand(g * g)
    outs = []
    for fused in (1, 0):
        p = Plan(op, 8, nvec=1)
        p.set_option(_lib.OPT_FUSED, fused)
        p.set_shifts([0.3])
        res = []
        for nu in (1, 2):
            p.upload(0, _lib.SLOT_V, 0, v0)
            p.upload(0, _lib.SLOT_F, 0, f)
            p.smooth(0, kind, nu, omega)
            res.append(p.download(0, _lib.SLOT_V, 0))
        p.upload(0, _lib.SLOT_V, 0, v0)
        p.upload(0, _lib.SLOT_F, 0, f)
        p.vcycle(2, 2, kind, omega=omega, nu_coarse=2)
        res.append(p.download(0, _lib.SLOT_V, 0))
        res.append(p.download(1, _lib.SLOT_F, 0))
        outs.append(res)
        p.close()
    for a, b in zip(*outs):
        assert rel_err(a, b) < 1e-11
    # and against the sparse oracle on the assembled matrix
    from oracle.sparse_ref import RefSolver, RefStencilMaker
    S, SM = RefSolver(), RefStencilMaker()
    smo = S.wjacobi if kind == _lib.WJACOBI else (lambda v, f, A, nu=4: S.gseidel_mc(v, f, A, nu=nu, dimension="2d"))
    y = S.vcycle(v0, f, op.tocsr(), SM, nu1=2, nu2=2, smoother=smo, shift=0.3, lowest_level=8, dimension="2d")
    # the oracle runs V(4,4) below the top level (reference quirk); rerun the device cycle the same way
    p = Plan(op, 8, nvec=1)
    p.set_shifts([0.3])
    p.upload(0, _lib.SLOT_V, 0, v0)
    p.upload(0, _lib.SLOT_F, 0, f)
    p.vcycle(2, 2, kind, omega=omega, nu_coarse=4)
    x = p.download(0, _lib.SLOT_V, 0)
    p.close()
    assert rel_err(x, y) < 1e-10


@pytest.mark.parametrize("kind,omega", [(_lib.WJACOBI, 2. / 3.), (_lib.GS_MC, 1.0)])
def test_recompute_instead_of_store_is_exact(backend, kind, omega):
    """MGCMT_OPT_RECOMPUTE: the down-leg pass skips storing the pre-smoothed iterate and the up-leg pass re-runs the
    same sweeps before adding the correction — the same arithmetic, so the cycle's result is bit-identical."""
    g = 256
    op = laplacian_operator(g, "2d") * SCALE
    rng = np.random.RandomState(21)
    v0, f = rng.rand(g * g), rng.rand(g * g)
    for nu1, nu2, nuc in ((2, 2, 2), (1, 2, 4), (3, 1, 2), (4, 4, 4)):
        outs = []
        for rec in (2, 0):                                  # 2 = on every fused level (default: only large ones)
            p = Plan(op, 8, nvec=1)
            p.set_option(_lib.OPT_RECOMPUTE, rec)
            p.set_shifts([0.6])
            p.upload(0, _lib.SLOT_V, 0, v0)
            p.upload(0, _lib.SLOT_F, 0, f)
            for _ in range(2):
                p.vcycle(nu1, nu2, kind, omega=omega, nu_coarse=nuc)
            outs.append(p.download(0, _lib.SLOT_V, 0))
            p.close()
        assert np.array_equal(outs[0], outs[1]), (nu1, nu2, nuc)


def test_potential_well_operator_three_terms(backend):
    """BASELINE config 5's operator: -laplacian/pi^2 + square-well potential = three Kronecker terms (Op9<3> on every
    level).  Fused vs one-launch-per-operation kernels, and the sparse oracle on the assembled matrix."""
    from multigridcmt_amd.operators import potential_well_operator
    from oracle.sparse_ref import RefSolver, RefStencilMaker
    g = 128
    op = potential_well_operator(g, depth=40.0, inner=(g // 4, 3 * g // 4))
    A = op.tocsr()
    ii, jj = np.meshgrid(np.arange(g), np.arange(g), indexing="ij")
    inside = (ii >= g // 4) & (ii < 3 * g // 4) & (jj >= g // 4) & (jj < 3 * g // 4)
    assert np.allclose(A.diagonal().reshape(g, g) - (4 * g * g / np.pi ** 2), np.where(inside, 0.0, 40.0))
    rng = np.random.RandomState(17)
    v0, f = rng.rand(g * g), rng.rand(g * g)
    S, SM = RefSolver(), RefStencilMaker()
    # the finest level is a 5-point operator with a product potential (two-colour Gauss-Seidel, recompute allowed),
    # its Galerkin coarsenings are general three-term 9-point operators (four colours)
    p = Plan(op, 8, nvec=1)
    assert p.fused_max_recompute(0, _lib.GS_MC, 2) == 2 and p.fused_max_recompute(1, _lib.GS_MC, 2) == 0
    p.close()
    for kind, omega, smo in ((_lib.WJACOBI, 2. / 3., S.wjacobi),
                             (_lib.GS_MC, 1.0, lambda v, f, A, nu=4: S.gseidel_mc(v, f, A, nu=nu, dimension="2d"))):
        outs = []
        for fused in (1, 0):
            p = Plan(op, 8, nvec=1)
            p.set_option(_lib.OPT_FUSED, fused)
            p.set_shifts([2.5])
            p.upload(0, _lib.SLOT_V, 0, v0)
            p.upload(0, _lib.SLOT_F, 0, f)
            p.vcycle(2, 2, kind, omega=omega, nu_coarse=4)
            outs.append(p.download(0, _lib.SLOT_V, 0))
            p.close()
        assert rel_err(outs[0], outs[1]) < 1e-11
        y = S.vcycle(v0, f, A, SM, nu1=2, nu2=2, smoother=smo, shift=2.5, lowest_level=8, dimension="2d")
        assert rel_err(outs[0], y) < 1e-10


@pytest.mark.parametrize("kind,omega", [(_lib.WJACOBI, 2. / 3.), (_lib.GS_MC, 1.0), (_lib.GS_MC, 1.25)])
def test_cycle_tail_in_one_launch(backend, kind, omega):
    """MGCMT_OPT_TAIL: the levels of at most 32 x 32 points (coarse solve included) as one LDS-resident launch give
    the cycle of the level-by-level launches — several vectors with their own shifts, coarsest grids 8 and 2, a cycle
    that starts right above the tail (64 x 64), and the three-term operator of a square well."""
    from multigridcmt_amd.operators import potential_well_operator
    rng = np.random.RandomState(3)
    cases = [(laplacian_operator(256, "2d") * SCALE, 8, 3), (laplacian_operator(128, "2d") * SCALE, 2, 1),
             (laplacian_operator(64, "2d") * SCALE, 4, 2), (potential_well_operator(128, 25.0, (40, 90)), 8, 1)]
    for op, lowest, k in cases:
        n = op.g * op.g
        v0, f = rng.rand(k, n), rng.rand(k, n)
        outs = []
        for tail in (1, 0):
            p = Plan(op, lowest, nvec=k)
            p.set_option(_lib.OPT_TAIL, tail)
            p.set_shifts(0.4 + 0.3 * np.arange(k))
            for q in range(k):
                p.upload(0, _lib.SLOT_V, q, v0[q])
                p.upload(0, _lib.SLOT_F, q, f[q])
            for _ in range(2):
                p.vcycle(2, 3, kind, omega=omega, k=k, nu_coarse=2)
            outs.append(np.stack([p.download(0, _lib.SLOT_V, q) for q in range(k)]))
            p.close()
        assert rel_err(outs[0], outs[1]) < 1e-12, (op.g, lowest, k)
