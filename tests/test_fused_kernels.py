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
    p.set_option(_lib.OPT_FUSED_ROWS, rows)      # rows a wave marches over (0 = automatic); a setting of this plan
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
@pytest.mark.parametrize("variable_laplacian", [False, True])
def test_fused_constant_part_plus_product_term(backend, kind, omega, variable_laplacian):
    """Scaled Laplacian plus a product potential p(i) q(j) (random, different in the two directions, non-zero on the last
    row and column).  variable_laplacian = False: the Laplacian terms coarsen to Toeplitz-but-last factors and the
    potential to ONE variable term — the Galerkin levels run the constant-part-plus-variable-term policy (Op9cv; the
    square well of BASELINE config 5 is this case).  True: the Laplacian terms carry potentials a(x), b(y) of their own,
    all three terms are variable — the general three-term policy (Op9<3>).  Fused against one-launch-per-operation
    kernels (smoothing passes on level 1, whole cycles with several chunk lengths, the restricted residual) and
    against the sparse oracle on the assembled matrix."""
    from multigridcmt_amd.operators import StructuredOperator, tri_identity, tri_laplacian
    from oracle.sparse_ref import RefSolver, RefStencilMaker
    g = 128
    rng = np.random.RandomState(23)
    Lx, Ly = tri_laplacian(g) * SCALE, tri_laplacian(g) * SCALE
    if variable_laplacian:
        Lx[1] += 3.0 * rng.rand(g)
        Ly[1] += 3.0 * rng.rand(g)
    dp, dq = np.zeros((3, g)), np.zeros((3, g))
    dp[1], dq[1] = 1.0 + 4.0 * rng.rand(g), 0.5 + 6.0 * rng.rand(g)
    op = StructuredOperator("2d", g, [(tri_identity(g), Ly), (Lx, tri_identity(g)), (dp, dq)])
    v0, f = rng.rand(g * g), rng.rand(g * g)
    v1, f1 = rng.rand(g * g // 4), rng.rand(g * g // 4)
    p = Plan(op, 8, nvec=1)
    kinds = [p.operator_kind(l) for l in range(3)]
    p.close()
    assert kinds == ([_lib.OPK_GENERAL] * 3 if variable_laplacian else [_lib.OPK_FIVE_DIAG, _lib.OPK_NINE_VAR, _lib.OPK_NINE_VAR])
    outs = []
    for fused, rows in ((1, 0), (1, 6), (1, 22), (0, 0)):
        p = Plan(op, 8, nvec=1)
        p.set_option(_lib.OPT_FUSED, fused)
        p.set_option(_lib.OPT_FUSED_ROWS, rows)
        p.set_shifts([0.45])
        res = []
        for nu in (1, 2, 3):
            p.upload(1, _lib.SLOT_V, 0, v1)
            p.upload(1, _lib.SLOT_F, 0, f1)
            p.smooth(1, kind, nu, omega)
            res.append(p.download(1, _lib.SLOT_V, 0))
        for nuc in (2, 1):
            p.upload(0, _lib.SLOT_V, 0, v0)
            p.upload(0, _lib.SLOT_F, 0, f)
            p.vcycle(2, 2, kind, omega=omega, nu_coarse=nuc)
            res.append(p.download(0, _lib.SLOT_V, 0))
            res.append(p.download(2, _lib.SLOT_F, 0))
        outs.append(res)
        p.close()
    for other in outs[:-1]:
        for a, b in zip(other, outs[-1]):
            assert rel_err(a, b) < 1e-11
    S, SM = RefSolver(), RefStencilMaker()
    smo = S.wjacobi if kind == _lib.WJACOBI else (lambda v, f, A, nu=4: S.gseidel_mc(v, f, A, nu=nu, dimension="2d"))
    y = S.vcycle(v0, f, op.tocsr(), SM, nu1=2, nu2=2, smoother=smo, shift=0.45, lowest_level=8, dimension="2d")
    p = Plan(op, 8, nvec=1)
    p.set_shifts([0.45])
    p.upload(0, _lib.SLOT_V, 0, v0)
    p.upload(0, _lib.SLOT_F, 0, f)
    p.vcycle(2, 2, kind, omega=omega, nu_coarse=4)                 # the oracle runs V(4,4) below the top level (reference quirk)
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


@pytest.mark.parametrize("kind,omega", [(_lib.WJACOBI, 2. / 3.), (_lib.GS_MC, 1.0), (_lib.GS_LEX, 1.0)])
def test_zero_start_flag_equals_cleared_iterate(backend, kind, omega):
    """MGCMT_CYCLE_ZERO_START (the reference's eigen-drivers start every cycle from zeros, 1DPotMatrixVcycle.py:70): the
    cycle takes "V is zero" as a flag — V, here full of garbage, is neither read nor cleared first — and gives the bits
    of a cycle on a cleared V; with and without recompute-instead-of-store, one and several columns, graph replay
    included (three calls), and from a coarser start level."""
    from multigridcmt_amd.operators import potential_well_operator
    for g, make in ((128, lambda g: laplacian_operator(g, "2d") * SCALE), (64, lambda g: potential_well_operator(g, 50.0, (g // 4, 3 * g // 4)))):
        op = make(g)
        rng = np.random.RandomState(5)
        for k, rec, level, nus in ((1, 2, 0, (2, 2, 2)), (2, 0, 0, (1, 3, 4)), (1, 2, 1, (2, 1, 2)), (1, 1, 0, (0, 2, 2))):
            n = (g >> level) ** 2
            f = rng.rand(k, n)
            outs = []
            for flagged in (True, False):
                p = Plan(op, 8, nvec=k)
                p.set_option(_lib.OPT_RECOMPUTE, rec)
                p.set_shifts(0.3 + 0.1 * np.arange(k))
                for q in range(k):
                    p.upload(level, _lib.SLOT_F, q, f[q])
                res = []
                for _ in range(3):
                    for q in range(k):
                        if flagged:
                            p.upload(level, _lib.SLOT_V, q, rng.rand(n) * 1e3)       # garbage the cycle must not see
                        else:
                            p.zero(level, _lib.SLOT_V, q)
                    p.vcycle(nus[0], nus[1], kind, omega=omega, k=k, nu_coarse=nus[2], level=level, zero_start=flagged)
                    res.append(np.stack([p.download(level, _lib.SLOT_V, q) for q in range(k)]))
                p.close()
                outs.append(res)
            for a, b in zip(*outs):
                assert np.array_equal(a, b), (g, k, rec, level, nus)


def test_marching_apply_against_assembled_matrix(backend):
    """mgcmt_apply on a finest 2-D level with a 5-point operator runs as a row march (k_apply_march: the scaled Laplacian
    of MGCMTStencilMaker.py:15-25 and the square-well Hamiltonian, a product potential on its diagonal) — against the
    assembled sparse matrix, with and without shifts, several columns, grids down to 2 x 2 and a chunk that ends inside
    the grid (40 rows)."""
    from multigridcmt_amd.operators import potential_well_operator
    rng = np.random.RandomState(11)
    for g, lowest in ((2, 2), (4, 2), (8, 8), (64, 8), (128, 8)):
        for make in (lambda g: laplacian_operator(g, "2d") * SCALE, lambda g: potential_well_operator(g, 37.0, (g // 4, 3 * g // 4))):
            op = make(g)
            A = op.tocsr()
            k = 3
            p = Plan(op, lowest, nvec=k)
            shifts = np.array([0.0, 0.7, -1.3])
            p.set_shifts(shifts)
            x = rng.rand(k, g * g) - 0.5
            for q in range(k):
                p.upload(0, _lib.SLOT_V, q, x[q])
            for q in range(k):
                for with_shift in (False, True):
                    p.apply(0, (_lib.SLOT_V, q), (_lib.SLOT_W, q), with_shift=with_shift)
                    want = A @ x[q] - (shifts[q] if with_shift else 0.0) * x[q]      # the shift of the source column
                    assert rel_err(p.download(0, _lib.SLOT_W, q), want) < 1e-14, (g, q, with_shift)
            rq, res = p.rayleigh_residual(0, _lib.SLOT_V, k)
            for q in range(k):
                ax = A @ x[q]
                assert abs(rq[q] - x[q] @ ax / (x[q] @ x[q])) < 1e-12 * abs(rq[q])
                assert abs(res[q] - np.linalg.norm(ax - shifts[q] * x[q])) < 1e-12 * res[q]
            p.close()


def test_ritz_pair_against_assembled_matrix(backend):
    """mgcmt_ritz_pair — <x,x>, <x,w>, <w,w>, <x,A w>, <w,A w> of rqmin's 2 x 2 pencil (MGCMTSolver.py:44-50) in one pass
    over x and w on 5-point levels (Laplacian, square well), composed from an application and a Gram pass elsewhere (level 1:
    9-point operators) — against the assembled matrices, vectors in different slots and columns, a grid with several row
    chunks per column block and one narrower than a wave."""
    from multigridcmt_amd.operators import potential_well_operator
    from multigridcmt_amd.stencil_maker import MGCMTStencilMaker
    SM = MGCMTStencilMaker()
    rng = np.random.RandomState(31)
    for g in (16, 128):
        for make in (lambda g: laplacian_operator(g, "2d") * SCALE, lambda g: potential_well_operator(g, 33.0, (g // 4, 3 * g // 4))):
            op = make(g)
            p = Plan(op, 8, nvec=2)
            p.set_shifts([0.7, -0.2])                         # the pencil uses the unshifted operator whatever the shifts
            A = op.tocsr()
            for level in (0, 1):
                gl = g >> level
                if level:
                    A = (SM.restriction(2 * gl, gl, dimension="2d") @ A @ SM.interpolation(gl, 2 * gl, dimension="2d")).tocsr()
                x, w = rng.rand(gl * gl) - 0.5, rng.rand(gl * gl) - 0.5
                X, W, S = (_lib.SLOT_W, 1), (_lib.SLOT_V, 0), (_lib.SLOT_W, 0)
                p.upload(level, X[0], X[1], x)
                p.upload(level, W[0], W[1], w)
                got = p.ritz_pair(level, X, W, S)
                aw = A @ w
                want = np.array([x @ x, x @ w, w @ w, x @ aw, w @ aw])
                assert np.all(np.abs(got - want) <= 1e-12 * np.abs(want).max()), (g, level, got, want)
            with pytest.raises(_lib.MgcmtError):
                p.ritz_pair(0, X, W, W)
            p.close()


def test_marching_apply_on_galerkin_levels(backend):
    """mgcmt_apply on the coarser levels (k_apply_march_terms: any sum of Kronecker terms) — operator A and mass operator
    M, the six applications per step and level of the reference's Rayleigh-quotient multigrid (MGCMTSolver.py:17-57,
    :99-122) — against R*A*P / R*M*P assembled from the stencil maker's matrices (MGCMTSolver.py:318) for a constant
    operator, the square well, an operator with three variable terms and the Mehrstellen pair."""
    from multigridcmt_amd.operators import (StructuredOperator, identity_operator, mehrstellen_mass, mehrstellen_operator,
                                            potential_well_operator, tri_identity, tri_laplacian)
    from multigridcmt_amd.stencil_maker import MGCMTStencilMaker
    SM = MGCMTStencilMaker()
    g = 64
    rng = np.random.RandomState(29)
    Lx, Ly = tri_laplacian(g) * SCALE, tri_laplacian(g) * SCALE
    Lx[1] += 3.0 * rng.rand(g)
    Ly[1] += 3.0 * rng.rand(g)
    dp, dq = np.zeros((3, g)), np.zeros((3, g))
    dp[1], dq[1] = 1.0 + rng.rand(g), 0.5 + rng.rand(g)
    cases = [(laplacian_operator(g, "2d") * SCALE, identity_operator(g, "2d"), _lib.OPK_NINE_CONST),
             (potential_well_operator(g, 30.0, (g // 4, 3 * g // 4)), identity_operator(g, "2d"), _lib.OPK_NINE_VAR),
             (StructuredOperator("2d", g, [(tri_identity(g), Ly), (Lx, tri_identity(g)), (dp, dq)]), identity_operator(g, "2d"), _lib.OPK_GENERAL),
             (mehrstellen_operator(g), mehrstellen_mass(g), _lib.OPK_NINE_CONST),
             # four terms (the library's maximum): variable Laplacian terms, a product potential and a cross term
             (StructuredOperator("2d", g, [(tri_identity(g), Ly), (Lx, tri_identity(g)), (dp, dq),
                                           (tri_laplacian(g) * (1.0 / (6.0 * g * g)), tri_laplacian(g) * SCALE)]), identity_operator(g, "2d"), _lib.OPK_GENERAL)]
    for op, mass, kind1 in cases:
        p = Plan(op, 8, nvec=2, mass=mass)
        p.set_shifts([0.0, 0.9])
        assert p.operator_kind(1) == kind1
        A, M = op.tocsr(), mass.tocsr()
        for level in (0, 1, 2):
            gl = g >> level
            if level > 0:
                R, P = SM.restriction(2 * gl, gl, dimension="2d"), SM.interpolation(gl, 2 * gl, dimension="2d")
                A, M = (R @ A @ P).tocsr(), (R @ M @ P).tocsr()
            x = rng.rand(2, gl * gl) - 0.5
            for q in range(2):
                p.upload(level, _lib.SLOT_V, q, x[q])
            for q in range(2):
                p.apply(level, (_lib.SLOT_V, q), (_lib.SLOT_W, q), with_shift=True)
                assert rel_err(p.download(level, _lib.SLOT_W, q), A @ x[q] - (0.9 if q else 0.0) * x[q]) < 1e-13, (kind1, level, q)
                p.apply(level, (_lib.SLOT_V, q), (_lib.SLOT_W, q), op=_lib.OP_M)
                assert rel_err(p.download(level, _lib.SLOT_W, q), M @ x[q]) < 1e-13, (kind1, level, q)
        p.close()


def test_potential_well_operator_three_terms(backend):
    """BASELINE config 5's operator: -laplacian/pi^2 + square-well potential = three Kronecker terms (a 5-point operator
    with a product potential on the finest level, a constant 9-point part plus one variable term — Op9cv — below).  Fused vs one-launch-per-operation kernels, and the sparse oracle on the assembled matrix."""
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
    # its Galerkin coarsenings are 9-point operators with a constant part and one variable term (four colours)
    p = Plan(op, 8, nvec=1)
    assert p.fused_max_recompute(0, _lib.GS_MC, 2) == 2 and p.fused_max_recompute(1, _lib.GS_MC, 2) == 0
    assert [p.operator_kind(l) for l in range(4)] == [_lib.OPK_FIVE_DIAG] + [_lib.OPK_NINE_VAR] * 3
    p.close()
    p = Plan(laplacian_operator(g, "2d") * SCALE, 8, nvec=1)
    assert [p.operator_kind(l) for l in range(3)] == [_lib.OPK_FIVE_POINT, _lib.OPK_NINE_CONST, _lib.OPK_NINE_CONST]
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
    """MGCMT_OPT_TAIL: the levels of at most 32 x 32 points (coarse solve included) as one launch — the dense product with
    the tail's matrix (1, the default) and the LDS-resident launch of ~45 phases (2) — give the cycle of the
    level-by-level launches (0): several vectors with their own shifts, coarsest grids 8 and 2, a cycle that starts right
    above the tail (64 x 64), the three-term operator of a square well, and a change of shifts between two cycles (the
    matrix is redone, also behind a replayed graph)."""
    from multigridcmt_amd.operators import potential_well_operator
    rng = np.random.RandomState(3)
    cases = [(laplacian_operator(256, "2d") * SCALE, 8, 3), (laplacian_operator(128, "2d") * SCALE, 2, 1),
             (laplacian_operator(64, "2d") * SCALE, 4, 2), (potential_well_operator(128, 25.0, (40, 90)), 8, 1)]
    if backend == "emu":        # (forming the matrix of a 32 x 32 tail is 1024 emulated workgroups per vector: one case, one vector)
        cases = [(laplacian_operator(64, "2d") * SCALE, 8, 1)] if kind == _lib.WJACOBI else [(laplacian_operator(32, "2d") * SCALE, 4, 2)]
    for op, lowest, k in cases:
        n = op.g * op.g
        v0, f = rng.rand(k, n), rng.rand(k, n)
        outs = []
        for tail in (1, 2, 0):
            p = Plan(op, lowest, nvec=k)
            p.set_option(_lib.OPT_TAIL, tail)
            p.set_shifts(0.4 + 0.3 * np.arange(k))
            for q in range(k):
                p.upload(0, _lib.SLOT_V, q, v0[q])
                p.upload(0, _lib.SLOT_F, q, f[q])
            for _ in range(3):
                p.vcycle(2, 3, kind, omega=omega, k=k, nu_coarse=2)
            p.set_shifts(0.1 + 0.2 * np.arange(k))          # other shifts: the same (possibly replayed) cycle, another matrix
            for _ in range(2):
                p.vcycle(2, 3, kind, omega=omega, k=k, nu_coarse=2)
            outs.append(np.stack([p.download(0, _lib.SLOT_V, q) for q in range(k)]))
            p.close()
        assert rel_err(outs[0], outs[2]) < 1e-12, (op.g, lowest, k, "dense")
        assert rel_err(outs[1], outs[2]) < 1e-12, (op.g, lowest, k, "phases")


@pytest.mark.parametrize("k", [2, 6, 10, 12])
def test_blocked_gram_schmidt_of_long_columns(backend, k):
    """MGCMT_OPT_MGS_BLOCK: the modified Gram-Schmidt of MGCMTProcessor.py:44-50 on columns too long for one workgroup as two
    passes over the data (Gram matrix, its Cholesky factor R, Q = A R^-1 — the same Q in exact arithmetic).  Well-conditioned
    columns (multigrid iterates are nearly orthonormal from the cycle before): equal to the column-by-column kernels and to
    the CPU restatement to rounding, orthonormal to rounding, column lengths of any size.  Ill-conditioned columns (the epsilon
    vectors of UnitTests/GramSchmidt.py stretched to long columns, cond 1e8): the gate on the device sends them through the
    column-by-column kernels — bit for bit what the option switched off computes."""
    from oracle.sparse_ref import RefProcessor
    from multigridcmt_amd.operators import laplacian_operator
    g = 128                                            # 16384 points: more than one workgroup's worth
    n = g * g
    rng = np.random.RandomState(3 + k)
    Q0 = np.linalg.qr(rng.rand(n, k))[0]
    mix = np.eye(k) + 0.2 * rng.rand(k, k)             # cond ~ 2-4
    A = (Q0 @ mix) * (10.0 ** rng.randint(-3, 4, size=k))[None, :]      # columns of very different lengths
    eps = 1e-8
    L = np.zeros((n, k))
    L[0, :] = 1.0
    for j in range(k):
        L[1 + j::k + 1, j] = eps                       # Laeuchli-type: pairwise nearly parallel
    p = Plan(laplacian_operator(g, "2d"), 8, nvec=k)
    results = {}
    for name, cols in (("well", A), ("ill", L)):
        for opt in (1, 0):
            p.set_option(_lib.OPT_MGS_BLOCK, opt)
            for q in range(k):
                p.upload(0, _lib.SLOT_V, q, cols[:, q])
            p.gramschmidt(0, _lib.SLOT_V, k, modified=1)
            results[(name, opt)] = np.stack([p.download(0, _lib.SLOT_V, q) for q in range(k)], axis=1)
    p.close()
    want = RefProcessor().gramschmidt(A, modified=1)
    assert np.abs(results[("well", 1)] - results[("well", 0)]).max() < 5e-14
    assert np.abs(results[("well", 1)] - want).max() < 5e-14
    assert np.abs(results[("well", 1)].T @ results[("well", 1)] - np.eye(k)).max() < 1e-13
    assert np.array_equal(results[("ill", 1)], results[("ill", 0)])
    assert np.abs(results[("ill", 0)] - RefProcessor().gramschmidt(L, modified=1)).max() < 1e-6    # (cond 1e8: summation orders show at cond * eps)
