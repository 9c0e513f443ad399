"""Parity of the product path (MGCMTSolver / MGCMTStencilMaker / MGCMTProcessor -> ctypes -> C-ABI ->
HIP kernels) with the reference.

Written like the reference's own UnitTests/*.py: same inputs, same calls, the expected numbers from
their comments; then the golden vectors produced by running the reference (oracle/gen_golden.py).
Tolerance: the north star's 1e-10 relative on fp64 results; most cases agree to ~1e-14 and are held
to 1e-12 so that a real regression is visible.
"""
import functools
import warnings

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import load_golden, rel_err
from multigridcmt_amd import MGCMTProcessor, MGCMTSolver, MGCMTStencilMaker

TIGHT = 1e-12
NORTH_STAR = 1e-10


@pytest.fixture
def trio(backend):
    return MGCMTSolver(), MGCMTStencilMaker(), MGCMTProcessor()


def H(sm, g, dim="1d"):
    return (-1 / np.pi ** 2) * sm.laplacian(g, dimension=dim)


# ---- UnitTests/wjacobiTest.py, gseidelTest.py, sorTest.py, vcycleTest.py, twogridTest.py -------------

def test_wjacobi_unit_test(trio):
    solver, stencil_maker, _ = trio
    gridsize = 2 ** 4
    A = stencil_maker.laplacian(gridsize)
    f = np.zeros((gridsize, 1))
    x = np.ones((gridsize, 1))
    for _ in range(5):
        x = solver.wjacobi(x, f, A, nu=4)
    assert x.shape == (gridsize, 1)
    assert abs(np.linalg.norm(x) - 2.94959) < 6e-6                  # UnitTests/wjacobiTest.py:25
    assert rel_err(x.ravel(), load_golden("kat_unit_tests")["wjacobi_x"]) < TIGHT


def test_gseidel_unit_test(trio):
    solver, stencil_maker, _ = trio
    gridsize = 2 ** 4
    A = stencil_maker.laplacian(gridsize)
    f = np.zeros((gridsize, 1))
    x = np.ones((gridsize, 1))
    for _ in range(5):
        x = solver.gseidel(x, f, A)
    assert abs(np.linalg.norm(x) - 1.88358) < 6e-6                  # UnitTests/gseidelTest.py:25
    assert rel_err(x.ravel(), load_golden("kat_unit_tests")["gseidel_x"]) < TIGHT


def test_sor_unit_test(trio):
    solver, stencil_maker, _ = trio
    gridsize = 2 ** 4
    A = stencil_maker.laplacian(gridsize)
    f = np.zeros((gridsize, 1))
    x = np.ones((gridsize, 1))
    for _ in range(5):
        x = solver.sor(x, f, A, nu=4, omega=2. / 3.)
    assert abs(np.linalg.norm(x) - 2.63327) < 6e-6                  # UnitTests/sorTest.py:25
    assert rel_err(x.ravel(), load_golden("kat_unit_tests")["sor_x"]) < TIGHT


def test_vcycle_unit_test(trio):
    solver, stencil_maker, _ = trio
    gridsize = 2 ** 4
    A = stencil_maker.laplacian(gridsize)
    x = solver.vcycle(np.ones((gridsize, 1)), np.zeros((gridsize, 1)), A, stencil_maker, nu1=4, nu2=4)
    assert x.shape == (gridsize,)                                   # MGCMTSolver.py:329
    assert abs(np.linalg.norm(x) - 0.17756) < 6e-6                  # UnitTests/vcycleTest.py:25
    assert rel_err(x, load_golden("kat_unit_tests")["vcycle_x"]) < TIGHT


def test_twogrid_unit_test(trio):
    solver, stencil_maker, _ = trio
    gridsize = 2 ** 4
    A = stencil_maker.laplacian(gridsize)
    x = solver.twogrid(np.ones((gridsize, 1)), np.zeros((gridsize, 1)), A, stencil_maker, nu1=4, nu2=4)
    assert abs(np.linalg.norm(x) - 0.04979) < 6e-6                  # UnitTests/twogridTest.py:25
    assert rel_err(x, load_golden("kat_unit_tests")["twogrid_x"]) < TIGHT


def test_vcycle_matrix_unit_test(trio):
    """UnitTests/vcycle_matrixTest.py:10-39 (its third block is stale: it predates the Gram-Schmidt
    inside vcycle_matrix and calls it with the crashing shifts=None default)."""
    solver, stencil_maker, _ = trio
    gold = load_golden("kat_vcycle_matrix_n4")
    n, num_vectors = 2 ** 2, 3
    laplacian = stencil_maker.laplacian(n)
    f_matrix = np.zeros((n, num_vectors))
    for i in range(num_vectors):
        f_matrix[:, i] = np.ones((n, 1))[0] * i
    x_single = np.ones((n, num_vectors)) * 4
    x_twogrid = np.ones((n, num_vectors)) * 4
    for i in range(num_vectors):
        x_single[:, i] = solver.vcycle(x_single[:, i].copy(), f_matrix[:, i].copy(), laplacian, stencil_maker)
        x_twogrid[:, i] = solver.twogrid(x_twogrid[:, i].copy(), f_matrix[:, i].copy(), laplacian, stencil_maker)
    expected = [-0.382301639189, -0.0257586075778, -0.840838733687]   # :27 (2nd value: misplaced decimal point there)
    for j in range(num_vectors):
        assert abs(np.dot(x_single[:, j].conj().T, laplacian.dot(x_single[:, j])) - expected[j]) < 1e-11
        assert abs(np.dot(x_twogrid[:, j].conj().T, laplacian.dot(x_twogrid[:, j])) - expected[j]) < 1e-11
    assert rel_err(x_single, gold["x_vcycle"]) < TIGHT
    x_matrix = solver.vcycle_matrix(np.ones((n, num_vectors)) * 4, f_matrix, laplacian, stencil_maker, shifts=np.zeros(3))
    assert rel_err(x_matrix[:, :2], gold["x_vcycle_matrix"][:, :2]) < NORTH_STAR   # 3rd column is rounding noise


# ---- golden vectors from the reference -------------------------------------------------------------

@pytest.mark.parametrize("tag,dim,g", [("1d64", "1d", 64), ("2d16", "2d", 16)])
def test_smoothers_random(trio, tag, dim, g):
    solver, sm, _ = trio
    gold = load_golden("smoothers_random")
    A = H(sm, g, dim)
    A = A - sp.eye(A.shape[0]) * float(gold["shift"])          # a pre-shifted matrix, as vcycle hands to smoothers
    v0, f = gold[tag + "_v0"], gold[tag + "_f"]
    assert rel_err(solver.wjacobi(v0.copy(), f.copy(), A, nu=3, omega=0.8).ravel(), gold[tag + "_wjacobi"]) < TIGHT
    assert rel_err(solver.gseidel(v0.copy(), f.copy(), A, nu=3).ravel(), gold[tag + "_gseidel"]) < TIGHT
    assert rel_err(solver.sor(v0.copy(), f.copy(), A, nu=3, omega=1.5).ravel(), gold[tag + "_sor"]) < TIGHT


def test_vcycle_1d_golden(trio):
    solver, sm, _ = trio
    gold = load_golden("vcycle_1d")
    A = sm.laplacian(1024)
    # BASELINE config 1: 1-D Poisson 1024, two-level, Gauss-Seidel(4,4), f = 0, x0 = 1
    x = solver.twogrid(np.ones(1024), np.zeros(1024), A, sm, nu1=4, nu2=4, smoother=solver.gseidel)
    assert abs(np.linalg.norm(x) - 0.023328643431521426) < 1e-12
    assert rel_err(x, gold["cfg1_twogrid_gs"]) < NORTH_STAR
    x = solver.vcycle(np.ones(1024), np.zeros(1024), A, sm, nu1=4, nu2=4, smoother=solver.gseidel, lowest_level=512)
    assert rel_err(x, gold["cfg1_vcycle512_gs"]) < NORTH_STAR
    x = solver.vcycle(np.ones(1024), np.zeros(1024), A, sm, nu1=4, nu2=4, smoother=solver.gseidel)
    assert abs(np.linalg.norm(x) - 0.500560418530017) < 1e-11
    assert rel_err(x, gold["full1024_gs"]) < NORTH_STAR
    x = solver.twogrid(np.zeros(1024), gold["cfg1b_f"].copy(), A, sm, nu1=4, nu2=4, smoother=solver.gseidel)
    assert rel_err(x, gold["cfg1b_twogrid_gs"]) < NORTH_STAR
    A = H(sm, 128)
    for name, smo in (("wj", solver.wjacobi), ("gs", solver.gseidel), ("sor", functools.partial(solver.sor, omega=1.3))):
        x = solver.vcycle(np.zeros(128), gold["h128_f"].copy(), A, sm, nu1=2, nu2=3, smoother=smo, shift=0.9, lowest_level=8)
        assert rel_err(x, gold["h128_vcycle_%s_shift0.9_low8" % name]) < NORTH_STAR, name


def test_vcycle_2d_golden(trio):
    solver, sm, _ = trio
    gold = load_golden("vcycle_2d")
    for g in (16, 32):
        A = H(sm, g, "2d")
        f = gold["g%d_f" % g]
        for name, smo in (("wj", solver.wjacobi), ("gs", solver.gseidel)):
            x = solver.vcycle(np.zeros(g * g), f.copy(), A, sm, shift=1.9, smoother=smo, dimension="2d", lowest_level=8)
            assert rel_err(x, gold["g%d_%s_shift1.9_low8" % (g, name)]) < NORTH_STAR, (g, name)
        x = solver.vcycle(np.zeros(g * g), f.copy(), A, sm, nu1=2, nu2=2, dimension="2d")
        assert rel_err(x, gold["g%d_wj_v22_shift0_low2" % g]) < NORTH_STAR
    x = solver.vcycle(np.zeros(256), gold["g16_f"].copy(), H(sm, 16, "2d"), sm, shift=1.9, dimension="2d", lowest_level=8)
    assert abs(np.linalg.norm(x) - 37.530067463013886) < 1e-9          # SURVEY §8(c) loader-sanity value
    x = solver.vcycle(np.zeros(256), gold["g16_f"].copy(), sm.laplacian(16, "2d"), sm, nu1=3, nu2=1,
                      smoother=functools.partial(solver.sor, omega=1.2), dimension="2d", lowest_level=4)
    assert rel_err(x, gold["lap16_sor1.2_low4"]) < NORTH_STAR


def test_vcycle_redblack_equals_reference_with_injected_smoother(trio):
    """Performance-mode smoother: the reference's OWN vcycle with the multicolour smoother injected
    through its smoother= seam (fixture) vs the device V-cycle with gseidel_rb."""
    solver, sm, _ = trio
    gold = load_golden("vcycle_multicolour_injected")
    for dim, g in (("1d", 128), ("2d", 16), ("2d", 32)):
        A = H(sm, g, dim)
        tag = "%s_g%d" % (dim, g)
        x = solver.vcycle(np.zeros(A.shape[0]), gold[tag + "_f"].copy(), A, sm, nu1=2, nu2=2, smoother=solver.gseidel_rb,
                          shift=0.5, dimension=dim, lowest_level=4)
        assert rel_err(x, gold[tag + "_v22_shift0.5_low4"]) < NORTH_STAR, tag


def test_vcycle_matrix_golden(trio):
    solver, sm, _ = trio
    gold = load_golden("vcycle_matrix")
    x = solver.vcycle_matrix(np.zeros((64, 3)), gold["h64_F"], H(sm, 64), sm, shifts=gold["h64_shifts"], lowest_level=8)
    assert x.shape == (64, 3) and rel_err(x, gold["h64_wj_low8"]) < NORTH_STAR
    x = solver.vcycle_matrix(np.zeros((64, 3)), gold["h64_F"], H(sm, 64), sm, shifts=gold["h64_shifts"], lowest_level=8,
                             smoother=solver.gseidel)
    assert rel_err(x, gold["h64_gs_low8"]) < NORTH_STAR
    x = solver.vcycle_matrix(np.zeros((256, 3)), gold["h2d16_F"], H(sm, 16, "2d"), sm, shifts=gold["h2d16_shifts"],
                             lowest_level=4, dimension="2d")
    assert rel_err(x, gold["h2d16_wj_low4"]) < NORTH_STAR
    assert np.allclose(x.T @ x, np.eye(3), atol=1e-12)                 # Gram-Schmidt on the way up (:434)


def test_gramschmidt_unit_test(trio):
    """UnitTests/GramSchmidt.py:10-129 — expected vectors / inner products from its comments."""
    _, _, processor = trio
    gold = load_golden("gramschmidt")
    machine_eps = np.finfo(float).eps
    A = np.column_stack((np.array([1, machine_eps, machine_eps]), np.array([1, machine_eps, 0]), np.array([1, 0, machine_eps])))
    O = processor.gramschmidt(A, modified=0)
    assert abs(abs(np.inner(O[:, 1], O[:, 2])) - 0.707) < 1e-3       # :44 CGS loses orthogonality
    O1 = processor.gramschmidt(A, modified=1)
    assert np.allclose(O1[:, 1], [0, 0, -1], atol=1e-15) and np.allclose(O1[:, 2], [0, -1, 0], atol=1e-15)   # :55-56
    assert abs(np.inner(O1[:, 1], O1[:, 2])) < 1e-15                 # :62
    for key in ("G1", "G2", "G3"):
        assert np.allclose(processor.gramschmidt(gold[key], modified=0), gold[key + "_cgs"], rtol=0, atol=1e-13)
        assert np.allclose(processor.gramschmidt(gold[key], modified=1), gold[key + "_mgs"], rtol=0, atol=1e-13)
    assert np.allclose(processor.normalize(gold["G3"]), gold["G3_normalize"], rtol=0, atol=1e-15)
    assert np.allclose(processor.orthogonality_check(gold["G3"]), gold["G3_gram"], rtol=0, atol=1e-12)
    v, u = gold["G3"][:, 0], gold["G3"][:, 1]
    assert np.allclose(processor.projection(v, u), (np.inner(v, u) / np.inner(u, u)) * u, rtol=0, atol=1e-14)


def test_rqmin_family(trio):
    """RQMin.py:15-50 — rqmin / vcycle_rqmg / vcycle_rqmg2 with M = I."""
    solver, sm, _ = trio
    gold = load_golden("rqmin")
    A, M = H(sm, 64), sp.eye(64)
    x, rho = solver.rqmin(A, gold["x0"], M, nu=4)
    assert abs(rho - 4.304377712544061) < 1e-10 * 4.3 and rel_err(x, gold["rqmin_x"]) < NORTH_STAR
    x = gold["x0"].copy()
    for i in range(2):
        x, rho = solver.vcycle_rqmg(x, A, M)
        assert abs(rho - gold["rqmg_rhos"][i]) < NORTH_STAR * abs(gold["rqmg_rhos"][i])
    assert abs(rho - 0.9706044628745033) < 1e-10
    assert rel_err(x, gold["rqmg_x"]) < 1e-9
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        X = solver.vcycle_rqmg2(gold["X0"], H(sm, 32), sp.eye(32), nmin=4)
    A32 = H(sm, 32)
    rq = [X[:, i] @ (A32 @ X[:, i]) / (X[:, i] @ X[:, i]) for i in range(2)]
    assert np.allclose(rq, [0.963017800746613, 1.040538684849706], rtol=1e-9, atol=0)
    assert np.allclose(rq, gold["rqmg2_nmin4_rq"], rtol=NORTH_STAR, atol=0)
    # the vectors themselves (the sign of a column is the sign of an eigenvector of the 2x2 pencil, :48 — free)
    for i in range(2):
        ref_col = gold["rqmg2_nmin4_X"][:, i]
        sign = np.sign(np.dot(X[:, i], ref_col))
        assert rel_err(sign * X[:, i], ref_col) < 1e-8, i
    with pytest.raises(AttributeError):
        solver.rqmin(A, gold["x0"])                                    # M=None crashes in the reference too (:19)
    with pytest.raises(NameError):
        solver.twogridrqmin(A, gold["x0"], M)                          # dead code in the reference (:167)


def test_rqmin_family_with_a_mass_operator(trio):
    """rqmin / vcycle_rqmg / vcycle_rqmg2 with a NON-IDENTITY mass operator, M = the finite-element mass matrix
    tridiag(1, 4, 1) / 6 (the reference coarsens M alongside A, MGCMTSolver.py:78-79,110-111) — reference outputs in
    tests/golden/rqmin_mass.npz.  On the device every level applies M in the passes of csrc/kernels_rq.hip and takes
    <g, M g> from one more application."""
    solver, sm, _ = trio
    gold = load_golden("rqmin_mass")

    def mass(n):
        return sp.diags([np.full(n - 1, 1.0 / 6.0), np.full(n, 4.0 / 6.0), np.full(n - 1, 1.0 / 6.0)], [-1, 0, 1], format="csr")

    A, M = H(sm, 64), mass(64)
    x, rho = solver.rqmin(A, gold["x0"], M, nu=4)
    assert abs(rho - float(gold["rqmin_rho"])) < NORTH_STAR * abs(float(gold["rqmin_rho"])) and rel_err(x, gold["rqmin_x"]) < NORTH_STAR
    x = gold["x0"].copy()
    for i in range(2):
        x, rho = solver.vcycle_rqmg(x, A, M)
        assert abs(rho - gold["rqmg_rhos"][i]) < NORTH_STAR * abs(gold["rqmg_rhos"][i])
    assert rel_err(x, gold["rqmg_x"]) < 1e-9
    x, rho = solver.twogridrqmin(A, gold["x0"], M, repaired=True)
    assert abs(rho - float(gold["twogrid_rho"])) < 1e-9 and rel_err(x, gold["twogrid_x"]) < 1e-8
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        X = solver.vcycle_rqmg2(gold["X0"], H(sm, 32), mass(32), nmin=4)
    A32, M32 = H(sm, 32), mass(32)
    rq = [X[:, i] @ (A32 @ X[:, i]) / (X[:, i] @ (M32 @ X[:, i])) for i in range(2)]
    assert np.allclose(rq, gold["rqmg2_nmin4_rq"], rtol=1e-9, atol=0)
    for i in range(2):
        ref_col = gold["rqmg2_nmin4_X"][:, i]
        sign = np.sign(np.dot(X[:, i], ref_col))
        assert rel_err(sign * X[:, i], ref_col) < 1e-8, i


def test_repaired_rayleigh_quotient_variants(trio):
    """SURVEY §8 (f)4.  twogridrqmin(repaired=True) is the reference's own Rayleigh-quotient multigrid cut off after one
    coarsening: pinned by the reference's vcycle_rqmg(..., nmin=n/2) (fixture rqmin.npz: twogrid_*).  vcycle_rqmg2 with
    the reference's DEFAULT nmin=2 dies inside eig in the reference (degenerate 2x2 pencil); repaired=True runs it:
    parity unpinned (nothing to compare with) — checked by what it must do: finite columns, orthonormal after the
    Gram-Schmidt passes of level 0, Rayleigh quotients no worse than the nmin=4 run the reference can do."""
    solver, sm, _ = trio
    gold = load_golden("rqmin")
    A, M = H(sm, 64), sp.eye(64)
    x, rho = solver.twogridrqmin(A, gold["x0"], M, repaired=True)
    assert abs(rho - float(gold["twogrid_rho"])) < NORTH_STAR * abs(float(gold["twogrid_rho"]))
    assert rel_err(np.sign(np.dot(x, gold["twogrid_x"])) * x, gold["twogrid_x"]) < 1e-9
    A32 = H(sm, 32)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        X2 = solver.vcycle_rqmg2(gold["X0"], A32, sp.eye(32), repaired=True)               # nmin=2, the default
        X4 = solver.vcycle_rqmg2(gold["X0"], A32, sp.eye(32), nmin=4)
    assert np.all(np.isfinite(X2))
    rq = lambda X: np.array([X[:, i] @ (A32 @ X[:, i]) / (X[:, i] @ X[:, i]) for i in range(2)])
    assert np.all(rq(X2) < rq(X4) * 1.05) and rq(X2)[0] < 1.0


@pytest.mark.parametrize("name,dim,g,low", [("driver_1dpot_matrix_vcycle", "1d", 128, 16),
                                            ("driver_2dpot_matrix_vcycle", "2d", 32, 4)])
def test_driver_reenactment(trio, name, dim, g, low):
    """The outer shift-and-invert loop of 1DPotMatrixVcycle.py:68-75 / 2DPotMatrixVcycle.py:91-105:
    Rayleigh-quotient eigenvalues after every iteration within 1e-10 relative of the reference's."""
    solver, sm, _ = trio
    gold = load_golden(name)
    A = H(sm, g, dim)
    V = gold["V0"].copy()
    k = V.shape[1]
    for it in range(gold["rq_history"].shape[0]):
        w = solver.vcycle_matrix(np.zeros(V.shape), V, A, sm, shifts=gold["bad_vals"], lowest_level=low, dimension=dim)
        for j in range(k):
            V[:, j] = w[:, j] / np.linalg.norm(w[:, j])
        rq = np.array([np.dot(V[:, j].conj().T, A.dot(V[:, j])) for j in range(k)])
        assert np.allclose(rq, gold["rq_history"][it], rtol=NORTH_STAR, atol=0), (name, it)


# ---- the reference's quirks at the Python boundary --------------------------------------------------

def test_shape_and_error_conventions(trio, capsys):
    solver, sm, _ = trio
    A = sm.laplacian(8)
    v0, f = np.ones(8), np.zeros(8)
    out = solver.wjacobi(v0, f, A)
    assert out.shape == (8, 1) and v0.shape == (8, 1) and f.shape == (8, 1)     # in-place .shape (:187-191)
    v0, f = np.ones(8), np.zeros(8)
    out = solver.vcycle(v0, f, A, sm, lowest_level=8)
    assert out.shape == (8, 1) and v0.shape == (8, 1)                          # lowest level returns (n,1) (:305-308)
    assert rel_err(out.ravel(), np.zeros(8) + np.linalg.solve(A.toarray(), np.zeros(8))) == 0.0
    assert solver.vcycle(np.ones(1), np.zeros(1), sm.laplacian(1), sm) is None  # (:303-304)
    assert "Length of start vector is not a power of 2" in capsys.readouterr().out
    assert solver.vcycle(np.ones(12), np.zeros(12), sm.laplacian(12), sm) is None
    assert "power of 2" in capsys.readouterr().out
    with pytest.raises(TypeError):
        solver.vcycle(np.ones(8), np.zeros(8), A, sm, smoother="gseidel")


def test_seam_foreign_smoother_callable(trio):
    """Seam 1 (MGCMTSolver.py:313,326,416,432): ANY callable smoother(v, f, shifted_matrix, nu=) is honoured — applied on
    host arrays level by level, the transfers, the coarse solve and Gram-Schmidt on the device.  The fixtures below were
    written by the reference's own vcycle with exactly these lambda wrappers (oracle/gen_golden.py:135,156)."""
    solver, sm, _ = trio
    gold = load_golden("vcycle_1d")
    x = solver.vcycle(np.zeros(128), gold["h128_f"].copy(), H(sm, 128), sm, nu1=2, nu2=3,
                      smoother=lambda v, f, A, nu=4: solver.sor(v, f, A, nu=nu, omega=1.3), shift=0.9, lowest_level=8)
    assert x.shape == (128,) and rel_err(x, gold["h128_vcycle_sor_shift0.9_low8"]) < NORTH_STAR
    g2 = load_golden("vcycle_2d")
    x = solver.vcycle(np.zeros(256), g2["g16_f"].copy(), sm.laplacian(16, "2d"), sm, nu1=3, nu2=1,
                      smoother=lambda v, f, A, nu=4: solver.sor(v, f, A, nu=nu, omega=1.2), dimension="2d", lowest_level=4)
    assert rel_err(x, g2["lap16_sor1.2_low4"]) < NORTH_STAR
    # a smoother that owes nothing to this package (NumPy damped Jacobi on the scipy matrix it is handed): the cycle
    # must be the device cycle with the built-in wjacobi, also for k columns with Gram-Schmidt (vcycle_matrix)
    calls = []

    def numpy_jacobi(v, f, A, nu=4):
        calls.append((A.shape[0], nu))
        d = A.diagonal().reshape(-1, 1)
        v = np.array(v, dtype=float).reshape(-1, 1)
        for _ in range(nu):
            v = v + (2. / 3.) * (np.asarray(f).reshape(-1, 1) - A @ v) / d
        return v

    A2, f2 = H(sm, 32, "2d"), np.random.RandomState(6).rand(1024)
    a = solver.vcycle(np.zeros(1024), f2.copy(), A2, sm, nu1=2, nu2=1, smoother=numpy_jacobi, shift=1.1, dimension="2d", lowest_level=4)
    b = solver.vcycle(np.zeros(1024), f2.copy(), A2, sm, nu1=2, nu2=1, shift=1.1, dimension="2d", lowest_level=4)
    assert rel_err(a, b) < 1e-12
    assert calls == [(1024, 2), (256, 4), (64, 4), (64, 4), (256, 4), (1024, 1)]      # nu not forwarded below the top (:320)
    F = np.random.RandomState(7).rand(1024, 3)
    shifts = np.array([0.0, 0.5, 1.5])
    a = solver.vcycle_matrix(np.zeros((1024, 3)), F, A2, sm, shifts=shifts, smoother=numpy_jacobi, lowest_level=8, dimension="2d")
    b = solver.vcycle_matrix(np.zeros((1024, 3)), F, A2, sm, shifts=shifts, lowest_level=8, dimension="2d")
    assert rel_err(a, b) < 1e-11
    t = solver.twogrid(np.zeros(64), np.ones(64), H(sm, 64), sm, smoother=numpy_jacobi, shift=0.2)
    assert rel_err(t, solver.twogrid(np.zeros(64), np.ones(64), H(sm, 64), sm, shift=0.2)) < 1e-12


def test_seam_foreign_stencil_maker(trio):
    """Seam 2 (MGCMTSolver.py:310-311): a caller's stencil maker is accepted when its matrices are the built-in transfers
    (here the oracle's independent restatement) and REFUSED when they are not (injection instead of full weighting)."""
    from oracle.sparse_ref import RefStencilMaker
    solver, sm, _ = trio
    A, f = H(sm, 64), np.random.RandomState(1).rand(64)
    want = solver.vcycle(np.zeros(64), f.copy(), A, sm, lowest_level=8)
    assert np.array_equal(solver.vcycle(np.zeros(64), f.copy(), A, RefStencilMaker(), lowest_level=8), want)

    class Injection(RefStencilMaker):
        def restriction(self, old, new, dimension="1d"):
            R = super().restriction(old, new, dimension=dimension).tolil()
            R[:, :] = 0
            for i in range(R.shape[0]):
                R[i, 2 * i + 1] = 1.0
            return R.tocsr()

    with pytest.raises(ValueError, match="differ from full weighting"):
        solver.vcycle(np.zeros(64), f.copy(), A, Injection(), lowest_level=8)
    with pytest.raises(ValueError):
        solver.vcycle_matrix(np.zeros((64, 2)), np.ones((64, 2)), A, Injection(), shifts=np.zeros(2), lowest_level=8)


def test_nu_not_forwarded_to_coarse_levels(trio):
    """MGCMTSolver.py:320 — coarse levels always run V(4,4): nu_coarse=4 (default) must differ from
    a uniform V(1,1) and equal the golden (covered above); uniform cycle available via nu_coarse."""
    solver, sm, _ = trio
    A = sm.laplacian(64)
    f = np.random.RandomState(2).rand(64)
    a = solver.vcycle(np.zeros(64), f.copy(), A, sm, nu1=1, nu2=1)
    b = solver.vcycle(np.zeros(64), f.copy(), A, sm, nu1=1, nu2=1, nu_coarse=1)
    assert rel_err(a, b) > 1e-6


def test_interpolate_restrict_match_stencil_maker_matrices(trio):
    solver, sm, _ = trio
    rng = np.random.RandomState(4)
    for dim, old, new in (("1d", 4, 16), ("1d", 8, 16), ("2d", 4, 16), ("2d", 8, 16)):
        c = rng.rand(old if dim == "1d" else old * old)
        assert rel_err(solver.interpolate(c, sm, new, dimension=dim), sm.interpolation(old, new, dimension=dim) * c) < 1e-14
        fv = rng.rand(new if dim == "1d" else new * new)
        assert rel_err(solver.restrict(fv, sm, old, dimension=dim), sm.restriction(new, old, dimension=dim) * fv) < 1e-14


def test_rayleigh_quotient_multigrid_two_dimensional(trio):
    """BASELINE config 5 at a size the oracle can run: Rayleigh-quotient multigrid for the ground state of a 2-D
    square well, M = I.  The reference's vcycle_rqmg only has 1-D transfers (MGCMTSolver.py:107-108); the oracle's
    dimension="2d" variant is the same algorithm with the 2-D ones (parity unpinned by the reference itself)."""
    from multigridcmt_amd.operators import identity_operator, potential_well_operator
    from oracle.sparse_ref import RefSolver
    solver, sm, _ = trio
    g = 32
    op = potential_well_operator(g, depth=30.0, inner=(8, 24))
    A, M = op.tocsr(), sp.eye(g * g, format="csr")
    x0 = np.random.RandomState(0).rand(g * g)
    S = RefSolver()
    x, xr = x0.copy(), x0.copy()
    for _ in range(2):
        x, rho = solver.vcycle_rqmg(x, op, identity_operator(g, "2d"), nu1=3, nu2=3, nmin=4)
        xr, rho_ref = S.vcycle_rqmg(xr, A, M, nu1=3, nu2=3, nmin=4, dimension="2d")
        assert abs(rho - np.real(rho_ref)) < NORTH_STAR * abs(np.real(rho_ref))
    assert rel_err(x, np.real(xr)) < 1e-8
    # and it is an eigenvalue estimate from above that two cycles bring close to the lowest eigenvalue
    import scipy.sparse.linalg as sla
    lowest = sla.eigsh(A, k=1, which="SA")[0][0]
    assert lowest <= rho < lowest * 1.05


def test_square_well_hamiltonian_against_reference(trio):
    """BASELINE config 5 at a size the reference runs (tests/golden/potential_well_2d.npz, written by running the
    reference's own vcycle / rqmin on the assembled sparse Hamiltonian): the same sparse matrix handed to the product
    is recognised as three Kronecker terms (Op9<3> kernels on every level) and must give the reference's numbers."""
    solver, sm, _ = trio
    gold = load_golden("potential_well_2d")
    g, depth, (lo, hi) = int(gold["g"]), float(gold["depth"]), gold["inner"]
    chi = np.zeros(g)
    chi[lo:hi] = 1.0
    H = ((-1 / np.pi ** 2) * sm.laplacian(g, dimension="2d") + sp.diags(depth * (1.0 - np.outer(chi, chi)).reshape(-1))).tocsr()
    from multigridcmt_amd.operators import potential_well_operator, recognise
    assert len(recognise(H).terms) == 3
    assert abs(recognise(H).tocsr() - H).max() < 1e-12
    assert abs(potential_well_operator(g, depth, (lo, hi)).tocsr() - H).max() < 1e-12
    for name, smo in (("wj", solver.wjacobi), ("gs", solver.gseidel)):
        w = solver.vcycle(gold["x0"].copy(), gold["f"].copy(), H, sm, nu1=2, nu2=2, shift=0.7, smoother=smo, dimension="2d",
                          lowest_level=8)
        assert rel_err(w, gold["vcycle_%s_v22_shift0.7_low8" % name]) < NORTH_STAR, name
    x, rho = solver.rqmin(H, gold["x0"].copy(), sp.eye(g * g), nu=6)
    assert abs(rho - float(gold["rqmin_rho"])) < NORTH_STAR * abs(float(gold["rqmin_rho"]))
    assert rel_err(x, gold["rqmin_x"]) < 1e-8


@pytest.mark.parametrize("case", ["1d", "1d_mass", "2d_well", "2d_mass"])
def test_rqmin_single_launch_equals_the_passes(trio, case, monkeypatch):
    """Levels of at most 4096 points take the whole rqmin call in one single-workgroup launch (kernels_rq.hip:
    k_rq_small); MGCMT_RQ_SMALL=0 sends them through the two passes per step the big levels take.  The same arithmetic
    per point, other summation orders: agreement to rounding (amplified by the steps' conditioning), on the reference's
    own 1-D problem (RQMin.py:28-33), with a mass operator, and on 2-D levels."""
    from multigridcmt_amd.operators import potential_well_operator
    solver, sm, _ = trio
    if case.startswith("1d"):
        n = 64
        A = (-1 / np.pi ** 2) * sm.laplacian(n)
        M = sp.eye(n) if case == "1d" else sp.diags([np.full(n - 1, 1 / 6), np.full(n, 2 / 3), np.full(n - 1, 1 / 6)], [-1, 0, 1])
        x0 = np.random.RandomState(3).rand(n)
    else:
        g = 32
        A = potential_well_operator(g, depth=30.0, inner=(8, 24)).tocsr() if case == "2d_well" else (-1 / np.pi ** 2) * sm.laplacian(g, dimension="2d")
        m1 = sp.diags([np.full(g - 1, 1 / 6), np.full(g, 2 / 3), np.full(g - 1, 1 / 6)], [-1, 0, 1])
        M = sp.eye(g * g) if case == "2d_well" else sp.kron(m1, m1)
        x0 = np.random.RandomState(3).rand(g * g)
    out = {}
    for form in ("1", "0"):
        monkeypatch.setenv("MGCMT_RQ_SMALL", form)
        out[form] = solver.rqmin(A.tocsr(), x0.copy(), sp.csr_matrix(M), nu=5)
    (x1, rho1), (x0_, rho0) = out["1"], out["0"]
    assert abs(rho1 - rho0) < 1e-12 * abs(rho0)
    assert rel_err(x1, x0_) < 1e-10


def test_vcycle_matrix_on_levels_with_long_columns(trio, monkeypatch):
    """vcycle_matrix (MGCMTSolver.py:375-436) at a size whose upper levels orthonormalise their columns by the two-pass
    form of the modified Gram-Schmidt (csrc/kernels_blas.hip: Gram matrix, Cholesky factor, Q = A R^-1 — columns of more
    than 4096 points) against the CPU restatement, which orthonormalises column by column as MGCMTProcessor.py:44-50
    does: three outer iterations of the drivers' shift-and-invert loop (2DPotMatrixVcycle.py:91-105), 128^2, six columns."""
    from oracle.sparse_ref import RefSolver
    from multigridcmt_amd.plan import release_plans
    solver, sm, _ = trio
    monkeypatch.delenv("MGCMT_MGS_BLOCK_MIN", raising=False)   # (the tuning knob must not switch the form under test off)
    release_plans()
    g, k = 128, 6
    A = H(sm, g, "2d")
    modes = [(1, 1), (1, 2), (2, 1), (2, 2), (1, 3), (3, 1)]
    shifts = np.array([0.98 * (a * a + b * b) for a, b in modes])          # just below the eigenvalues a^2 + b^2
    rng = np.random.RandomState(4)
    V = rng.rand(g * g, k)
    Vr = V.copy()
    ref = RefSolver()
    for it in range(3):
        w = solver.vcycle_matrix(np.zeros(V.shape), V, A, sm, shifts=shifts, lowest_level=8, dimension="2d")
        wr = ref.vcycle_matrix(np.zeros(Vr.shape), Vr, A, sm, shifts=shifts, lowest_level=8, dimension="2d")
        assert rel_err(w, wr) < NORTH_STAR, it
        assert np.abs(w.T @ w - np.eye(k)).max() < 1e-12
        V, Vr = w, wr
    release_plans()
