"""The NumPy oracle (oracle/sparse_ref.py) against the reference's known answers and golden vectors.

Golden vectors come from running the reference itself (oracle/gen_golden.py).  Tolerance: 1e-12
relative for well-conditioned cases (both sides are fp64 with different summation order), looser
where a shift makes (A - mu I) ill-conditioned — stated per test.
"""
import warnings

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import load_golden, rel_err
from oracle.sparse_ref import RefProcessor, RefSolver, RefStencilMaker

S, SM, P = RefSolver(), RefStencilMaker(), RefProcessor()
H = lambda g, dim="1d": (-1 / np.pi ** 2) * SM.laplacian(g, dimension=dim)


def test_unit_test_known_answers():
    """UnitTests/{wjacobi,gseidel,sor,vcycle,twogrid}Test.py:22-25 — printed norms, 5-6 digits."""
    g = load_golden("kat_unit_tests")
    n = 16
    A = SM.laplacian(n)
    f = np.zeros(n)
    outs = []
    x = np.ones(n)
    for _ in range(5):
        x = S.wjacobi(x, f, A, nu=4)
    outs.append(("wjacobi_x", x))
    x = np.ones(n)
    for _ in range(5):
        x = S.gseidel(x, f, A)
    outs.append(("gseidel_x", x))
    x = np.ones(n)
    for _ in range(5):
        x = S.sor(x, f, A, nu=4, omega=2. / 3.)
    outs.append(("sor_x", x))
    outs.append(("vcycle_x", S.vcycle(np.ones(n), f, A, SM, nu1=4, nu2=4)))
    outs.append(("twogrid_x", S.twogrid(np.ones(n), f, A, SM, nu1=4, nu2=4)))
    for (key, x), expected in zip(outs, g["expected_norms"]):
        assert abs(np.linalg.norm(x) - expected) < 6e-6, key       # the comment's printed digits
        assert rel_err(x, g[key]) < 1e-12, key                     # the reference's full vector


def test_vcycle_matrix_test_n4():
    """UnitTests/vcycle_matrixTest.py:21-39."""
    g = load_golden("kat_vcycle_matrix_n4")
    lap = SM.laplacian(4)
    F = g["F"]
    for i in range(3):
        xv = S.vcycle(np.ones(4) * 4, F[:, i], lap, SM)
        xt = S.twogrid(np.ones(4) * 4, F[:, i], lap, SM)
        assert rel_err(xv, g["x_vcycle"][:, i]) < 1e-12
        assert rel_err(xt, g["x_twogrid"][:, i]) < 1e-12
        assert abs(xv @ (lap @ xv) - g["quad_comment"][i]) < 1e-11
    xm = S.vcycle_matrix(np.ones((4, 3)) * 4, F, lap, SM, shifts=np.zeros(3))
    # columns 0 and 1 only: the three right-hand sides are multiples of one vector, so after two
    # Gram-Schmidt projections the third column is pure rounding noise in the reference too
    assert rel_err(xm[:, :2], g["x_vcycle_matrix"][:, :2]) < 1e-10


def test_operators():
    g = load_golden("operators")
    assert np.array_equal(SM.restriction(16, 8).toarray(), g["R_16_8"])
    assert np.array_equal(SM.interpolation(8, 16).toarray(), g["P_8_16"])
    assert np.array_equal(SM.interpolation(4, 16).toarray(), g["P_4_16"])
    assert np.array_equal(SM.restriction(16, 4).toarray(), g["R_16_4"])
    assert np.array_equal(SM.interpolation(4, 8, dimension="2d").toarray(), g["P2d_4_8"])
    assert np.array_equal(SM.restriction(8, 4, dimension="2d").toarray(), g["R2d_8_4"])
    assert np.array_equal(SM.restriction(16, 4, dimension="2d").toarray(), g["R2d_16_4"])
    assert np.array_equal(SM.interpolation(4, 16, dimension="2d").toarray(), g["P2d_4_16"])
    assert np.array_equal(SM.laplacian(8).toarray(), g["L1d_8"])
    assert np.array_equal(SM.laplacian(4, dimension="2d").toarray(), g["L2d_4"])
    rap = (SM.restriction(16, 8) @ SM.laplacian(16) @ SM.interpolation(8, 16)).toarray()
    assert np.allclose(rap, g["RAP_16"], rtol=0, atol=1e-12)


def test_operator_error_convention(capsys):
    """MGCMTStencilMaker.py:44-49,69-74 — message printed, None returned."""
    assert SM.interpolation(8, 4) is None
    assert SM.interpolation(6, 16) is None
    assert SM.interpolation(4, 12) is None
    assert SM.restriction(4, 8) is None
    out = capsys.readouterr().out
    assert "isn't" in out


@pytest.mark.parametrize("tag,dim,g", [("1d64", "1d", 64), ("2d16", "2d", 16)])
def test_smoothers_random(tag, dim, g):
    gold = load_golden("smoothers_random")
    A = H(g, dim)
    A = A - sp.eye(A.shape[0]) * float(gold["shift"])
    v0, f = gold[tag + "_v0"], gold[tag + "_f"]
    assert rel_err(S.wjacobi(v0, f, A, nu=3, omega=0.8), gold[tag + "_wjacobi"]) < 1e-12
    assert rel_err(S.gseidel(v0, f, A, nu=3), gold[tag + "_gseidel"]) < 1e-12
    assert rel_err(S.sor(v0, f, A, nu=3, omega=1.5), gold[tag + "_sor"]) < 1e-12


def test_vcycle_1d():
    gold = load_golden("vcycle_1d")
    A = SM.laplacian(1024)
    one, zero = np.ones(1024), np.zeros(1024)
    x = S.twogrid(one, zero, A, SM, nu1=4, nu2=4, smoother=S.gseidel)
    assert rel_err(x, gold["cfg1_twogrid_gs"]) < 1e-10
    assert abs(np.linalg.norm(x) - float(gold["cfg1_norm_survey"])) < 1e-12
    x = S.vcycle(one, zero, A, SM, nu1=4, nu2=4, smoother=S.gseidel, lowest_level=512)
    assert rel_err(x, gold["cfg1_vcycle512_gs"]) < 1e-10
    x = S.vcycle(one, zero, A, SM, nu1=4, nu2=4, smoother=S.gseidel)
    assert rel_err(x, gold["full1024_gs"]) < 1e-10
    assert abs(np.linalg.norm(x) - float(gold["full1024_norm_survey"])) < 1e-11
    x = S.twogrid(zero, gold["cfg1b_f"], A, SM, nu1=4, nu2=4, smoother=S.gseidel)
    assert rel_err(x, gold["cfg1b_twogrid_gs"]) < 1e-10
    A = H(128)
    f = gold["h128_f"]
    sor13 = lambda v, f, A, nu=4: S.sor(v, f, A, nu=nu, omega=1.3)
    for name, smo in (("wj", S.wjacobi), ("gs", S.gseidel), ("sor", sor13)):
        x = S.vcycle(np.zeros(128), f, A, SM, nu1=2, nu2=3, smoother=smo, shift=0.9, lowest_level=8)
        assert rel_err(x, gold["h128_vcycle_%s_shift0.9_low8" % name]) < 1e-10, name


def test_vcycle_2d():
    gold = load_golden("vcycle_2d")
    for g in (16, 32):
        A = H(g, "2d")
        f = gold["g%d_f" % g]
        for name, smo in (("wj", S.wjacobi), ("gs", S.gseidel)):
            x = S.vcycle(np.zeros(g * g), f, A, SM, shift=1.9, smoother=smo, dimension="2d", lowest_level=8)
            assert rel_err(x, gold["g%d_%s_shift1.9_low8" % (g, name)]) < 1e-10, (g, name)
            if g == 16 and name == "wj":
                assert abs(np.linalg.norm(x) - float(gold["g16_norm_survey"])) < 1e-9
        x = S.vcycle(np.zeros(g * g), f, A, SM, nu1=2, nu2=2, dimension="2d")
        assert rel_err(x, gold["g%d_wj_v22_shift0_low2" % g]) < 1e-10
    sor12 = lambda v, f, A, nu=4: S.sor(v, f, A, nu=nu, omega=1.2)
    x = S.vcycle(np.zeros(256), gold["g16_f"], SM.laplacian(16, "2d"), SM, nu1=3, nu2=1, smoother=sor12,
                 dimension="2d", lowest_level=4)
    assert rel_err(x, gold["lap16_sor1.2_low4"]) < 1e-10


def test_vcycle_multicolour_through_reference_seam():
    """The reference's own vcycle with the multicolour smoother injected through smoother= (the
    performance-mode oracle) equals the restatement's vcycle with the same smoother."""
    gold = load_golden("vcycle_multicolour_injected")
    for dim, g in (("1d", 128), ("2d", 16), ("2d", 32)):
        A = H(g, dim)
        tag = "%s_g%d" % (dim, g)
        smo = lambda v, f, A, nu=4, d=dim: S.gseidel_mc(v, f, A, nu=nu, dimension=d)
        x = S.vcycle(np.zeros(A.shape[0]), gold[tag + "_f"], A, SM, nu1=2, nu2=2, smoother=smo, shift=0.5,
                     dimension=dim, lowest_level=4)
        assert rel_err(x, gold[tag + "_v22_shift0.5_low4"]) < 1e-10, tag


def test_vcycle_matrix():
    gold = load_golden("vcycle_matrix")
    x = S.vcycle_matrix(np.zeros((64, 3)), gold["h64_F"], H(64), SM, shifts=gold["h64_shifts"], lowest_level=8)
    assert rel_err(x, gold["h64_wj_low8"]) < 1e-9
    x = S.vcycle_matrix(np.zeros((64, 3)), gold["h64_F"], H(64), SM, shifts=gold["h64_shifts"], lowest_level=8,
                        smoother=S.gseidel)
    assert rel_err(x, gold["h64_gs_low8"]) < 1e-9
    x = S.vcycle_matrix(np.zeros((256, 3)), gold["h2d16_F"], H(16, "2d"), SM, shifts=gold["h2d16_shifts"],
                        lowest_level=4, dimension="2d")
    assert rel_err(x, gold["h2d16_wj_low4"]) < 1e-9


def test_gramschmidt():
    """UnitTests/GramSchmidt.py:10-67,73-129 — expected vectors in the comments (:32-44,54-67)."""
    gold = load_golden("gramschmidt")
    mgs = P.gramschmidt(gold["G1"], modified=1)
    assert np.allclose(mgs[:, 1], [0, 0, -1], atol=1e-15) and np.allclose(mgs[:, 2], [0, -1, 0], atol=1e-15)
    cgs = P.gramschmidt(gold["G1"], modified=0)
    assert abs(abs(np.inner(cgs[:, 1], cgs[:, 2])) - 0.7071067811865) < 1e-6        # CGS loses orthogonality
    for key in ("G1", "G2", "G3"):
        assert np.allclose(P.gramschmidt(gold[key], modified=0), gold[key + "_cgs"], rtol=0, atol=1e-13)
        assert np.allclose(P.gramschmidt(gold[key], modified=1), gold[key + "_mgs"], rtol=0, atol=1e-13)
    assert np.allclose(P.normalize(gold["G3"]), gold["G3_normalize"], rtol=0, atol=1e-15)
    assert np.allclose(P.orthogonality_check(gold["G3"]), gold["G3_gram"], rtol=0, atol=1e-12)


def test_rqmin_family():
    gold = load_golden("rqmin")
    A, M = H(64), sp.eye(64)
    x, rho = S.rqmin(A, gold["x0"], M, nu=4)
    assert abs(np.real(rho) - float(gold["rqmin_rho_survey"])) < 1e-10
    assert rel_err(np.real(x), gold["rqmin_x"]) < 1e-10
    x = gold["x0"].copy()
    for i in range(2):
        x, rho = S.vcycle_rqmg(x, A, M)
        assert abs(np.real(rho) - gold["rqmg_rhos"][i]) < 1e-10
    assert abs(np.real(rho) - float(gold["rqmg_rho_survey"])) < 1e-10
    assert rel_err(np.real(x), gold["rqmg_x"]) < 1e-9
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        X = S.vcycle_rqmg2(gold["X0"], H(32), sp.eye(32), nmin=4)
    rq = [X[:, i] @ (H(32) @ X[:, i]) / (X[:, i] @ X[:, i]) for i in range(2)]
    assert np.allclose(rq, gold["rqmg2_nmin4_rq_survey"], rtol=0, atol=1e-9)


def _fem_mass(n):
    return sp.diags([np.full(n - 1, 1.0 / 6.0), np.full(n, 4.0 / 6.0), np.full(n - 1, 1.0 / 6.0)], [-1, 0, 1], format="csr")


def test_rqmin_family_with_a_mass_operator():
    """The same with M = the finite-element mass matrix tridiag(1, 4, 1) / 6: the reference coarsens M alongside A
    (MGCMTSolver.py:78-79,110-111); fixture rqmin_mass.npz is the reference's own output."""
    gold = load_golden("rqmin_mass")
    A, M = H(64), _fem_mass(64)
    x, rho = S.rqmin(A, gold["x0"], M, nu=4)
    assert abs(np.real(rho) - float(gold["rqmin_rho"])) < 1e-10 and rel_err(np.real(x), gold["rqmin_x"]) < 1e-10
    x = gold["x0"].copy()
    for i in range(2):
        x, rho = S.vcycle_rqmg(x, A, M)
        assert abs(np.real(rho) - gold["rqmg_rhos"][i]) < 1e-10
    assert rel_err(np.real(x), gold["rqmg_x"]) < 1e-9
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        X = S.vcycle_rqmg2(gold["X0"], H(32), _fem_mass(32), nmin=4)
    rq = [X[:, i] @ (H(32) @ X[:, i]) / (X[:, i] @ (_fem_mass(32) @ X[:, i])) for i in range(2)]
    assert np.allclose(rq, gold["rqmg2_nmin4_rq"], rtol=0, atol=1e-9)


def test_driver_reenactments():
    """Compute sections of 1DPotMatrixVcycle.py:42-80 and 2DPotMatrixVcycle.py:54-109 (reduced)."""
    for name, dim, g, low in (("driver_1dpot_matrix_vcycle", "1d", 128, 16), ("driver_2dpot_matrix_vcycle", "2d", 32, 4)):
        gold = load_golden(name)
        A = H(g, dim)
        V = gold["V0"].copy()
        k = V.shape[1]
        for it in range(gold["rq_history"].shape[0]):
            w = S.vcycle_matrix(np.zeros(V.shape), V, A, SM, shifts=gold["bad_vals"], lowest_level=low, dimension=dim)
            for j in range(k):
                V[:, j] = w[:, j] / np.linalg.norm(w[:, j])
            rq = np.array([V[:, j] @ (A @ V[:, j]) for j in range(k)])
            # Rayleigh quotients: 1e-10 relative (north_star); shifts sit next to eigenvalues
            assert np.allclose(rq, gold["rq_history"][it], rtol=1e-10, atol=0), (name, it)


def test_oracle_square_well_hamiltonian():
    """The sparse oracle on the reference's outputs for the 2-D square-well Hamiltonian (config 5)."""
    gold = load_golden("potential_well_2d")
    g, depth, (lo, hi) = int(gold["g"]), float(gold["depth"]), gold["inner"]
    S, SM = RefSolver(), RefStencilMaker()
    chi = np.zeros(g)
    chi[lo:hi] = 1.0
    H = ((-1 / np.pi ** 2) * SM.laplacian(g, dimension="2d") + sp.diags(depth * (1.0 - np.outer(chi, chi)).reshape(-1))).tocsr()
    for name, smo in (("wj", S.wjacobi), ("gs", S.gseidel)):
        w = S.vcycle(gold["x0"], gold["f"], H, SM, nu1=2, nu2=2, smoother=smo, shift=0.7, lowest_level=8, dimension="2d")
        assert rel_err(w, gold["vcycle_%s_v22_shift0.7_low8" % name]) < 1e-11, name
    x, rho = S.rqmin(H, gold["x0"], sp.eye(g * g), nu=6)
    assert abs(np.real(rho) - float(gold["rqmin_rho"])) < 1e-11 * abs(float(gold["rqmin_rho"]))
    assert rel_err(np.real(x), gold["rqmin_x"]) < 1e-9
