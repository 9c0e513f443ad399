"""The matrix-free C oracle (oracle/mgcmt_oracle.c) against the generic-sparse NumPy oracle, which is
itself pinned by the reference's golden vectors.  Runs on CPU."""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import load_golden, rel_err
from oracle import structured as st
from oracle.sparse_ref import RefSolver, RefStencilMaker

S, SM = RefSolver(), RefStencilMaker()


def test_galerkin_factors_match_sparse_triple_product():
    for n in (8, 16, 64):
        L = SM.laplacian(n)
        R, P = SM.restriction(n, n // 2), SM.interpolation(n // 2, n)
        rap = (R @ L @ P).toarray()
        c = st.galerkin(st.tri_laplacian(n))
        assert np.allclose(np.diag(rap), c[1], rtol=1e-14) and np.allclose(np.diag(rap, -1), c[0, 1:], rtol=1e-14)
        assert np.allclose(np.diag(rap, 1), c[2, :-1], rtol=1e-14)
        t = st.galerkin(st.tri_identity(n))                       # R1 P1 = tridiag(1/8, 3/4, 1/8), last 5/8
        assert np.allclose((R @ P).toarray().diagonal(), t[1]) and t[1, -1] == 0.625 and t[1, 0] == 0.75


@pytest.mark.parametrize("dim,g", [("1d", 64), ("2d", 16), ("2d", 32)])
def test_level_operations(dim, g):
    scale, shift = -1 / np.pi ** 2, 1.3
    X, Y = st.laplacian_factors(g, dim, scale)
    A = scale * SM.laplacian(g, dimension=dim)
    n = A.shape[0]
    Ash = A - shift * sp.eye(n)
    rng = np.random.RandomState(9)
    v, f = rng.rand(n), rng.rand(n)
    assert rel_err(st.apply(X, Y, shift, v), Ash @ v) < 1e-13
    assert rel_err(st.residual(X, Y, shift, v, f), f - Ash @ v) < 1e-13
    assert rel_err(st.smooth(X, Y, shift, st.WJACOBI, v, f, 3, 0.8), S.wjacobi(v, f, Ash, nu=3, omega=0.8)) < 1e-13
    assert rel_err(st.smooth(X, Y, shift, st.GS_LEX, v, f, 3), S.gseidel(v, f, Ash, nu=3)) < 1e-13
    assert rel_err(st.smooth(X, Y, shift, st.SOR_LEX, v, f, 3, 1.5), S.sor(v, f, Ash, nu=3, omega=1.5)) < 1e-13
    assert rel_err(st.smooth(X, Y, shift, st.GS_MC, v, f, 2, 1.0), S.gseidel_mc(v, f, Ash, nu=2, dimension=dim)) < 1e-13
    d = 1 if dim == "1d" else 2
    nr = 1 if dim == "1d" else g
    assert rel_err(st.restrict(d, nr, g, v), SM.restriction(g, g // 2, dimension=dim) @ v) < 1e-14
    c = rng.rand((g // 2) ** d)
    assert rel_err(st.prolong(d, nr, g, c), SM.interpolation(g // 2, g, dimension=dim) @ c) < 1e-14


@pytest.mark.parametrize("dim,g,low", [("1d", 256, 2), ("2d", 32, 8), ("2d", 64, 4)])
def test_vcycle_matches_sparse_oracle(dim, g, low):
    scale, shift = -1 / np.pi ** 2, 0.7
    X, Y = st.laplacian_factors(g, dim, scale)
    A = scale * SM.laplacian(g, dimension=dim)
    f = np.random.RandomState(3).rand(A.shape[0])
    z = np.zeros(A.shape[0])
    for kind, smo in ((st.WJACOBI, S.wjacobi), (st.GS_LEX, S.gseidel),
                      (st.GS_MC, lambda v, f, A, nu=4: S.gseidel_mc(v, f, A, nu=nu, dimension=dim))):
        omega = 2. / 3. if kind == st.WJACOBI else 1.0
        x = st.vcycle(X, Y, g, low, shift, kind, z, f, 2, 3, 4, omega)
        y = S.vcycle(z, f, A, SM, nu1=2, nu2=3, smoother=smo, shift=shift, lowest_level=low, dimension=dim)
        assert rel_err(x, y) < 1e-11, kind


def test_vcycle_against_reference_golden():
    gold = load_golden("vcycle_2d")
    X, Y = st.laplacian_factors(32, "2d", -1 / np.pi ** 2)
    x = st.vcycle(X, Y, 32, 8, 1.9, st.WJACOBI, np.zeros(1024), gold["g32_f"], 4, 4, 4, 2. / 3.)
    assert rel_err(x, gold["g32_wj_shift1.9_low8"]) < 1e-10
    x = st.vcycle(X, Y, 32, 8, 1.9, st.GS_LEX, np.zeros(1024), gold["g32_f"], 4, 4, 4, 1.0)
    assert rel_err(x, gold["g32_gs_shift1.9_low8"]) < 1e-10
