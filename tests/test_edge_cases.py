"""Edge cases of the boundary: zero sweeps, start grid == lowest level, many columns, strided column views,
ragged / non-power-of-two sizes, 2-D twogrid, plan cache reuse, and the sparse oracle as the checker."""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import rel_err
from multigridcmt_amd import MGCMTProcessor, MGCMTSolver, MGCMTStencilMaker
from oracle.sparse_ref import RefProcessor, RefSolver, RefStencilMaker

S, SM = RefSolver(), RefStencilMaker()


@pytest.fixture
def trio(backend):
    return MGCMTSolver(), MGCMTStencilMaker(), MGCMTProcessor()


def H(sm, g, dim="1d"):
    return (-1 / np.pi ** 2) * sm.laplacian(g, dimension=dim)


def test_zero_sweeps_and_single_sweeps(trio):
    solver, sm, _ = trio
    rng = np.random.RandomState(0)
    for dim, g in (("1d", 64), ("2d", 16)):
        A = H(sm, g, dim)
        n = A.shape[0]
        f, v0 = rng.rand(n), rng.rand(n)
        for nu1, nu2 in ((0, 0), (1, 0), (0, 1), (1, 1), (5, 3)):
            x = solver.vcycle(v0.copy(), f.copy(), A, sm, nu1=nu1, nu2=nu2, shift=0.3, dimension=dim, lowest_level=4)
            y = S.vcycle(v0, f, A, SM, nu1=nu1, nu2=nu2, shift=0.3, dimension=dim, lowest_level=4)
            assert rel_err(x, y) < 1e-11, (dim, nu1, nu2)
        assert rel_err(solver.wjacobi(v0.copy(), f.copy(), A, nu=0).ravel(), v0) == 0.0


def test_start_grid_is_lowest_level(trio):
    solver, sm, _ = trio
    for dim, g in (("1d", 32), ("2d", 8)):
        A = H(sm, g, dim)
        n = A.shape[0]
        f = np.random.RandomState(1).rand(n)
        x = solver.vcycle(np.zeros(n), f.copy(), A, sm, shift=0.7, dimension=dim, lowest_level=g)
        assert x.shape == (n, 1)
        assert rel_err(x.ravel(), sp.linalg.spsolve(sp.csc_matrix(A - 0.7 * sp.eye(n)), f)) < 1e-11


def test_many_columns_and_strided_views(trio):
    solver, sm, processor = trio
    rng = np.random.RandomState(2)
    n, k = 256, 32                                     # the maximum number of columns of one call
    A = H(sm, n)
    F = rng.rand(n, k)
    shifts = np.linspace(0.1, 0.9, k)
    # lowest level 32: the smallest grid that is Gram-Schmidt'ed has 64 points (> k columns)
    x = solver.vcycle_matrix(np.zeros((n, k)), F, A, sm, shifts=shifts, lowest_level=32)
    y = S.vcycle_matrix(np.zeros((n, k)), F, A, SM, shifts=shifts, lowest_level=32)
    assert rel_err(x, y) < 1e-8                        # 32 Gram-Schmidt'ed columns: conditioning, not kernels
    with pytest.raises(ValueError):
        solver.vcycle_matrix(np.zeros((n, 33)), rng.rand(n, 33), A, sm, shifts=np.zeros(33))
    # column views (stride k) as the reference's callers pass them (UnitTests/vcycle_matrixTest.py:23)
    X = rng.rand(n, 3)
    out = solver.vcycle(X[:, 1], F[:, 0], A, sm)
    assert rel_err(out, S.vcycle(X[:, 1].copy(), F[:, 0].copy(), A, SM)) < 1e-12
    G = processor.gramschmidt(F[:, :5])
    assert np.allclose(G, RefProcessor().gramschmidt(F[:, :5]), atol=1e-13)


def test_non_power_of_two_and_ragged(trio, capsys):
    solver, sm, _ = trio
    assert solver.vcycle(np.ones(24), np.zeros(24), sm.laplacian(24), sm) is None
    assert solver.vcycle(np.ones(36), np.zeros(36), sm.laplacian(6, "2d"), sm, dimension="2d") is None
    assert "power of 2" in capsys.readouterr().out
    with pytest.raises(ValueError):
        solver.vcycle(np.ones(16), np.zeros(16), sm.laplacian(16), sm, lowest_level=3)
    with pytest.raises(ValueError):
        solver.vcycle_matrix(np.zeros((16, 2)), np.zeros((16, 2)), sm.laplacian(16), sm, shifts=np.zeros(3))


def test_twogrid_two_dimensional(trio):
    """The reference's twogrid cannot run in 2-D (MGCMTSolver.py:350); here it equals a V-cycle whose lowest
    level is the next grid."""
    solver, sm, _ = trio
    A = H(sm, 16, "2d")
    f = np.random.RandomState(4).rand(256)
    a = solver.twogrid(np.zeros(256), f.copy(), A, sm, nu1=2, nu2=2, shift=0.4, dimension="2d")
    b = solver.vcycle(np.zeros(256), f.copy(), A, sm, nu1=2, nu2=2, shift=0.4, dimension="2d", lowest_level=8)
    assert rel_err(a, b) < 1e-13


def test_plan_cache_reuse_and_shift_changes(trio):
    """A driver's outer loop calls vcycle with the same operator and changing shifts: the cached plan must refactor
    its coarsest level when the shift changes."""
    solver, sm, _ = trio
    A = H(sm, 128)
    f = np.random.RandomState(5).rand(128)
    for mu in (0.0, 0.9, 0.9, 3.8, 0.0):
        x = solver.vcycle(np.zeros(128), f.copy(), A, sm, shift=mu, lowest_level=8)
        assert rel_err(x, S.vcycle(np.zeros(128), f, A, SM, shift=mu, lowest_level=8)) < 1e-10, mu


def test_one_dimensional_variable_coefficients(trio):
    """1-D operators are general tridiagonals (a potential on the diagonal, PotWellSolver.py:150-153 style)."""
    solver, sm, _ = trio
    n = 128
    V = np.where((np.arange(n) < 40) | (np.arange(n) >= 90), 25.0, 0.0)
    A = H(sm, n) + sp.diags(V)
    f = np.random.RandomState(6).rand(n)
    for smo, rsmo in ((solver.wjacobi, S.wjacobi), (solver.gseidel, S.gseidel),
                      (solver.gseidel_rb, lambda v, f, A, nu=4: S.gseidel_mc(v, f, A, nu=nu, dimension="1d"))):
        x = solver.vcycle(np.zeros(n), f.copy(), A, sm, nu1=2, nu2=2, smoother=smo, shift=1.1, lowest_level=4)
        y = S.vcycle(np.zeros(n), f, A, SM, nu1=2, nu2=2, smoother=rsmo, shift=1.1, lowest_level=4)
        assert rel_err(x, y) < 1e-10


def test_gram_matrix_and_linear_combinations(backend):
    """mgcmt_gram / mgcmt_lincomb (the fused vector algebra of the Rayleigh-Ritz steps) against NumPy, 1-D and 2-D,
    including a destination that is one of the inputs."""
    from multigridcmt_amd import _lib
    from multigridcmt_amd.operators import laplacian_operator
    from multigridcmt_amd.plan import Plan
    rng = np.random.RandomState(4)
    for op in (laplacian_operator(1024, "1d"), laplacian_operator(64, "2d")):
        p = Plan(op, 8, nvec=3)
        n = p.size(0)
        vs = rng.rand(6, n) - 0.5
        where = [(_lib.SLOT_V, 0), (_lib.SLOT_V, 1), (_lib.SLOT_F, 2), (_lib.SLOT_W, 0), (_lib.SLOT_W, 1), (_lib.SLOT_F, 0)]
        for v, (slot, q) in zip(vs, where):
            p.upload(0, slot, q, v)
        for nv in (1, 2, 4, 6):
            G = p.gram(0, where[:nv])
            assert np.allclose(G, vs[:nv] @ vs[:nv].T, rtol=1e-12, atol=1e-12)
            assert np.array_equal(G, G.T)
        p.lincomb(0, [(0.5, where[0]), (-2.0, where[1]), (3.0, where[2]), (0.25, where[3])], where[4])
        assert rel_err(p.download(0, *where[4]), 0.5 * vs[0] - 2.0 * vs[1] + 3.0 * vs[2] + 0.25 * vs[3]) < 1e-14
        p.lincomb(0, [(1.5, where[0]), (-1.0, where[5])], where[0])          # in place
        assert rel_err(p.download(0, *where[0]), 1.5 * vs[0] - vs[5]) < 1e-14
        p.close()


@pytest.mark.parametrize("dimension,g,lowest", [("1d", 256, 4), ("2d", 64, 8)])
def test_full_multigrid(trio, dimension, g, lowest):
    """MGCMTSolver.fmg (an addition; the reference only describes FMG in its report): equal to the oracle's FMG built
    from the reference-equivalent pieces, and one FMG pass leaves a far smaller residual than one V-cycle from zero."""
    from oracle.sparse_ref import RefSolver, RefStencilMaker
    solver, sm, _ = trio
    A = (-1 / np.pi ** 2) * sm.laplacian(g, dimension=dimension)
    n = A.shape[0]
    f = np.random.RandomState(8).rand(n)
    S, SM = RefSolver(), RefStencilMaker()
    Asp = (-1 / np.pi ** 2) * SM.laplacian(g, dimension=dimension)
    for smo, rsmo in ((solver.wjacobi, S.wjacobi), (solver.gseidel, S.gseidel)):
        w = solver.fmg(f.copy(), A, sm, nu1=2, nu2=2, smoother=smo, shift=0.3, lowest_level=lowest, dimension=dimension)
        ref = S.fmg(f, Asp, SM, nu1=2, nu2=2, smoother=rsmo, shift=0.3, lowest_level=lowest, dimension=dimension)
        assert rel_err(w, ref) < 1e-10
    shifted = Asp - 0.3 * sp.eye(n)
    one_cycle = solver.vcycle(np.zeros(n), f.copy(), A, sm, nu1=2, nu2=2, shift=0.3, lowest_level=lowest, dimension=dimension)
    w = solver.fmg(f.copy(), A, sm, nu1=2, nu2=2, shift=0.3, lowest_level=lowest, dimension=dimension)
    assert np.linalg.norm(f - shifted @ w) < 0.5 * np.linalg.norm(f - shifted @ one_cycle)     # (a rough random f)


def test_operator_mutated_in_place_between_calls(backend):
    """`A *= c` keeps id/shape/nnz; the second vcycle must solve with the NEW operator (oracle check)."""
    from multigridcmt_amd import MGCMTSolver, MGCMTStencilMaker
    from oracle.sparse_ref import RefSolver, RefStencilMaker
    solver, sm = MGCMTSolver(), MGCMTStencilMaker()
    ref, rsm = RefSolver(), RefStencilMaker()
    g = 32
    A = sp.csr_matrix((-1 / np.pi ** 2) * sm.laplacian(g, dimension="2d"))
    f = np.random.RandomState(4).rand(g * g)
    for factor in (1.0, 2.0, 0.25):
        A *= factor
        x = solver.vcycle(np.zeros(g * g), f.copy(), A, sm, nu1=2, nu2=2, shift=0.3, dimension="2d", lowest_level=8)
        y = ref.vcycle(np.zeros(g * g), f, A.copy(), rsm, nu1=2, nu2=2, shift=0.3, dimension="2d", lowest_level=8)
        assert rel_err(x, y) < 1e-11, factor


def test_block_gram_and_combine(backend):
    """mgcmt_block_gram / mgcmt_block_combine against numpy: up to 12 x 4 vectors, outputs aliasing inputs, bad arguments."""
    from multigridcmt_amd import _lib
    from multigridcmt_amd.operators import laplacian_operator
    from multigridcmt_amd.plan import Plan
    from multigridcmt_amd._lib import MgcmtError
    g, nv = 32, 12
    p = Plan(laplacian_operator(g, "2d"), 4, nvec=nv)
    rng = np.random.RandomState(2)
    host = {}
    for slot in (_lib.SLOT_W, _lib.SLOT_F):
        for j in range(nv):
            host[(slot, j)] = rng.randn(g * g)
            p.upload(0, slot, j, host[(slot, j)])
    A = [(_lib.SLOT_W, j) for j in range(12)]
    B = [(_lib.SLOT_F, j) for j in (3, 0, 7, 11)]
    G = p.block_gram(0, A, B)
    want = np.array([[host[a] @ host[b] for b in B] for a in A])
    assert np.allclose(G, want, rtol=1e-13, atol=1e-11)
    assert np.allclose(p.block_gram(0, A[:5], A[:3]), np.array([[host[a] @ host[b] for b in A[:3]] for a in A[:5]]), rtol=1e-13, atol=1e-11)
    C = rng.randn(7, 3)
    ins = A[:4] + B[:3]
    outs = [A[1], B[0], (_lib.SLOT_F, 5)]                   # two of the outputs are inputs
    p.block_combine(0, ins, outs, C)
    for j, o in enumerate(outs):
        assert np.allclose(p.download(0, o[0], o[1]), sum(C[i, j] * host[v] for i, v in enumerate(ins)), rtol=1e-13, atol=1e-12)
    with pytest.raises(MgcmtError):
        p.block_combine(0, ins, [A[1], A[1]], np.zeros((7, 2)))   # an output named twice
    with pytest.raises(MgcmtError):
        p.block_gram(0, A + [A[0]], B)                              # 13 vectors
    p.close()
