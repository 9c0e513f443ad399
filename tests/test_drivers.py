"""Counterparts of the reference's driver scripts (multigridcmt_amd/drivers.py) against closed-form eigenvalues of
the discrete box Hamiltonian (the reference's "integration test": exact values printed next to the multigrid ones,
1DPotMatrixVcycle.py:85-86, 2DPotMatrixVcycle.py:118-120) and against the numbers published in the reference's
report (AndyMartinez_MultigridExamen.pdf p.21-22 and p.44, quoted in BASELINE.md §1)."""
import numpy as np
import pytest

from conftest import load_golden, rel_err
from multigridcmt_amd import drivers


def test_exact_eigenvalue_formula():
    # n = 256, k = 1 -> 0.9922207 (PDF p.22)
    assert abs(drivers.exact_box_eigenvalues(256, "1d", 1)[0] - 0.9922207) < 1e-7
    assert np.allclose(drivers.exact_box_eigenvalues(64, "2d", 3), np.sort(np.add.outer(
        drivers.exact_box_eigenvalues(64, "1d", 4), drivers.exact_box_eigenvalues(64, "1d", 4)).ravel())[:3])


def test_1d_pot_matrix_vcycle_driver(backend):
    """1DPotMatrixVcycle.py with its own parameters, guesses taken from the fixture (ARPACK output)."""
    gold = load_golden("driver_1dpot_matrix_vcycle")
    out = drivers.shift_invert_eigenpairs("1d", 128, 16, 10, 10, guesses=(gold["bad_vals"], gold["bad_vecs"]))
    assert np.allclose(out["history"][1:], gold["rq_history"], rtol=1e-10, atol=0)      # the reference's own numbers
    exact = drivers.exact_box_eigenvalues(128, "1d", 10)
    assert np.allclose(out["eigenvalues"][:6], exact[:6], rtol=5e-5, atol=0)              # one V-cycle per step: ~1e-5, as in the PDF tables
    assert np.allclose(exact[:6], [1, 4, 9, 16, 25, 36], rtol=2e-2)                       # ... near the continuum n^2 (h = 1/128)


@pytest.mark.gpu
def test_2d_published_table(hip_only):
    """PDF p.44: 2-D box 128^2, guesses from 16^2, lowest V-cycle level 8 -> multigrid eigenvalues
    [1.96901685, 4.92197738, 4.92197786, 7.87499665, 9.84169385, 9.84166931] (Lanczos on the fine grid:
    1.96901511, 4.92195391 x2, 7.87489271, 9.84157268 x2).  Reproduced to the accuracy the method has."""
    out = drivers.shift_invert_eigenpairs("2d", 128, 16, 6, 5, lowest_level=8, tolerance=np.finfo(float).eps)
    published_mg = np.array([1.96901685, 4.92197738, 4.92197786, 7.87499665, 9.84169385, 9.84166931])
    lanczos = np.array([1.96901511, 4.92195391, 4.92195391, 7.87489271, 9.84157268, 9.84157268])
    got = np.sort(out["eigenvalues"])
    assert abs(got[0] - published_mg[0]) < 5e-9                     # the non-degenerate ground state: every printed digit
    assert np.allclose(got, np.sort(published_mg), rtol=0, atol=2e-4)   # degenerate pairs depend on ARPACK's rotation of the guesses
    assert np.allclose(got, lanczos, rtol=0, atol=4e-4)                 # five single-V-cycle steps: the method's own accuracy
    assert np.allclose(lanczos, drivers.exact_box_eigenvalues(128, "2d", 6), rtol=0, atol=1e-7)


def test_rqmin_driver(backend):
    """RQMin.py:28-50.  With the fixture's start vector (RandomState(0)) two vcycle_rqmg sweeps give the
    reference's own Rayleigh quotient (tests/golden/rqmin.npz); the second column is re-minimised and then deflated
    after every sweep, so its quotient (taken BEFORE the deflation, as the script does) falls back towards the
    ground state while the returned columns stay orthonormal."""
    gold = load_golden("rqmin")
    rho1, rho2, X = drivers.rayleigh_quotient_multigrid(64, 2, 10, seed=0)
    exact = drivers.exact_box_eigenvalues(64, "1d", 2)
    assert abs(rho1 - gold["rqmg_rhos"][1]) < 1e-10
    assert exact[0] <= rho1 < exact[0] + 2e-3 and exact[0] <= rho2 < exact[1]
    assert np.allclose(X.T @ X, np.eye(2), atol=1e-12)


def test_potential_well_eigensolve_two_dimensional(backend):
    """BASELINE config 5 at a CPU-checkable size: both the Rayleigh-quotient multigrid and the V-cycle-preconditioned
    Rayleigh-quotient minimisation converge to the lowest eigenvalue of the assembled sparse Hamiltonian (eigsh)."""
    import scipy.sparse.linalg as sla
    from multigridcmt_amd.operators import potential_well_operator
    g = 64
    lowest = sla.eigsh(potential_well_operator(g, 50.0, (16, 48)).tocsr(), k=1, which="SA")[0][0]
    hist = []
    rho, x = drivers.potential_well_eigensolve(g, depth=50.0, cycles=12, method="vcycle", nu=2, lowest=8, history=hist)
    assert all(b <= a + 1e-13 for a, b in zip(hist, hist[1:]))          # Rayleigh-Ritz never increases rho
    assert abs(rho - lowest) < 1e-9 * lowest
    H = potential_well_operator(g, 50.0, (16, 48)).tocsr()
    assert np.linalg.norm(H @ x - rho * x) < 1e-5 * np.linalg.norm(x)
    rho_mg, _ = drivers.potential_well_eigensolve(g, depth=50.0, cycles=4, method="rqmg", nu=4, lowest=4)
    assert lowest <= rho_mg < lowest * (1 + 1e-3)


@pytest.mark.parametrize("dimension,g,bad,fixture,iters,lowest", [("1d", 128, 16, "driver_1dpot_matrix_vcycle", 10, 16),
                                                                  ("2d", 32, 8, "driver_2dpot_matrix_vcycle", 3, 4)])
def test_resident_driver_equals_host_driver(backend, dimension, g, bad, fixture, iters, lowest):
    """The device-resident outer loop (guesses interpolated inside the plan, k-column cycles, normalisation and
    Rayleigh quotients without shuttling the columns over PCIe) computes what the host-array loop computes — which for
    these fixtures is what the reference's 1DPotMatrixVcycle.py / 2DPotMatrixVcycle.py compute."""
    gold = load_golden(fixture)
    k = gold["bad_vecs"].shape[1]
    guesses = (gold["bad_vals"], gold["bad_vecs"])
    stats = {}
    res = drivers.shift_invert_eigenpairs_resident(dimension, g, bad, k, iters, lowest_level=lowest, guesses=guesses, stats=stats)
    host = drivers.shift_invert_eigenpairs(dimension, g, bad, k, iters, lowest_level=lowest, guesses=guesses)
    assert np.allclose(res["history"], host["history"], rtol=1e-10, atol=0)
    assert np.allclose(res["history"][1:], gold["rq_history"], rtol=1e-10, atol=0)
    for j in range(k):
        assert rel_err(res["eigenvectors"][:, j], host["eigenvectors"][:, j]) < 1e-9
    assert stats["loop_seconds"] > 0


@pytest.mark.parametrize("placement", ["none", "after", "inside"])
def test_gram_schmidt_placements_of_1dpot_mgs(backend, placement):
    """1DPotMGS.py:50-127 at the script's own sizes: the Rayleigh-quotient history of each Gram-Schmidt placement,
    computed by the reference itself (tests/golden/driver_1dpot_mgs.npz), from the device-resident loop."""
    gold = load_golden("driver_1dpot_mgs")
    out = drivers.shift_invert_eigenpairs_resident("1d", 256, 8, 6, 10, lowest_level=8, guesses=(gold["bad_vals"], gold["bad_vecs"]),
                                                   gram_schmidt=placement)
    want = gold["rq_history_" + placement]
    assert np.allclose(out["history"], want, rtol=1e-10, atol=0), np.abs(out["history"] / want - 1).max()
    V, Vref = out["eigenvectors"], gold["V_final_" + placement]
    for j in range(6):
        sign = np.sign(np.dot(V[:, j], Vref[:, j]))
        assert rel_err(sign * V[:, j], Vref[:, j]) < 1e-8, (placement, j)


def test_residual_history_of_2dpot_matrix_vcycle(backend):
    """2DPotMatrixVcycle.py:100-101: ||(H - mu_i I) v_i|| per column and iteration, all columns from one batched pass."""
    gold = load_golden("driver_2dpot_matrix_vcycle")
    k = gold["bad_vecs"].shape[1]
    seen = []
    out = drivers.shift_invert_eigenpairs_resident("2d", 32, 8, k, 3, lowest_level=4, guesses=(gold["bad_vals"], gold["bad_vecs"]),
                                                   residuals=seen)
    assert np.allclose(out["residual_history"][1:], gold["residual_history"], rtol=1e-9, atol=1e-13)
    assert np.allclose(out["history"][1:], gold["rq_history"], rtol=1e-10, atol=0)
    assert len(seen) == 3 and np.array_equal(seen[-1], out["residual_history"][-1])


def test_nested_iteration_guesses(backend):
    """guess_method="fmg" (an addition; parity unpinned — the reference has no FMG code): the guesses it hands to the
    outer loop are better than straight interpolation of the coarse eigenvectors (smaller residuals at iteration 0) and
    the loop converges to the same eigenvalues."""
    k = 4
    plain = drivers.shift_invert_eigenpairs_resident("2d", 128, 16, k, 3, lowest_level=8, tolerance=1e-12)
    fmg = drivers.shift_invert_eigenpairs_resident("2d", 128, 16, k, 3, lowest_level=8, guess_method="fmg")
    exact = drivers.exact_box_eigenvalues(128, "2d", k)
    assert np.all(fmg["residual_history"][0] < plain["residual_history"][0])
    assert abs(np.sort(fmg["eigenvalues"])[0] - exact[0]) < 1e-5 and np.allclose(np.sort(fmg["eigenvalues"]), exact, atol=2e-3)


@pytest.mark.parametrize("use_p", [True, False])
def test_block_eigensolve_box(backend, use_p):
    """SURVEY par. 8(f)4 (not in the reference: parity unpinned): four lowest eigenpairs of -laplacian/pi^2 on 64^2 — with
    the degenerate pair (1,2)/(2,1) inside the block — against the exact discrete eigenvalues; the LOBPCG-style update
    converges faster than blocked steepest descent; the vectors come back orthonormal with small residuals."""
    from multigridcmt_amd.operators import laplacian_operator
    g, k = 64, 4
    op = laplacian_operator(g, "2d") * (-1 / np.pi ** 2)
    hist, res = [], []
    vals, vecs = drivers.block_eigensolve(op, k=k, cycles=16, lowest=4, history=hist, residuals=res, use_p=use_p)
    exact = drivers.exact_box_eigenvalues(g, "2d", k)
    # (the last vector of a block converges slowest: a factor 5 per iteration with P, 2.3 without)
    assert np.allclose(vals, exact, rtol=0, atol=1e-8 if use_p else 1e-3), np.abs(vals - exact)
    assert np.allclose(vals[:3], exact[:3], rtol=0, atol=1e-10 if use_p else 1e-6)
    assert np.abs(vecs.T @ vecs - np.eye(k)).max() < 1e-10
    A = op.tocsr()
    assert np.abs(A @ vecs - vecs * vals).max() < (1e-4 if use_p else 1e-1)
    assert all(np.all(np.diff(h) >= -1e-12) for h in hist)                 # Ritz values come sorted
    assert np.all(hist[-1] <= hist[0] + 1e-12) and res[-1].max() < res[0].max() * 1e-3


def test_block_eigensolve_square_well_against_eigsh(backend):
    """The same solver on BASELINE config 5's operator (2-D square well), 64^2, against scipy's eigsh."""
    import scipy.sparse.linalg as sla
    from multigridcmt_amd.operators import potential_well_operator
    g, k = 64, 3
    op = potential_well_operator(g, 50.0, (g // 4, 3 * g // 4))
    vals, vecs = drivers.block_eigensolve(op, k=k, cycles=10, lowest=8)
    want = np.sort(sla.eigsh(op.tocsr(), k=k, which="SA")[0])
    assert np.allclose(vals, want, rtol=1e-9, atol=1e-9)


def test_block_eigensolve_with_a_mass_operator(backend):
    """SURVEY par. 8(f)4, generalised: the lowest pairs of A x = lambda M x with A the compact fourth-order (Mehrstellen)
    9-point Laplacian and M its right-hand-side operator (operators.mehrstellen_mass) — the reference carries M through
    its Rayleigh-quotient routines (MGCMTSolver.py:33-50,78-79) — against scipy's eigsh on the assembled pencil; the
    eigenvalues are fourth-order accurate approximations of the continuum's (k^2 + l^2), which the 5-point operator with
    M = I is not; vectors come back M-orthonormal.  Not in the reference: parity unpinned."""
    import scipy.sparse.linalg as sla
    from multigridcmt_amd.operators import laplacian_operator, mehrstellen_mass, mehrstellen_operator
    g, k = 64, 3
    A, M = mehrstellen_operator(g) * (-1 / np.pi ** 2), mehrstellen_mass(g)
    hist, res = [], []
    vals, vecs = drivers.block_eigensolve(A, k=k, cycles=12, lowest=8, mass=M, history=hist, residuals=res)
    As, Ms = A.tocsr(), M.tocsr()
    want = np.sort(sla.eigsh(As, k=k, M=Ms, sigma=0.0, which="LM")[0])
    assert np.allclose(vals, want, rtol=1e-9, atol=1e-9), np.abs(vals - want)
    assert np.abs(vecs.T @ (Ms @ vecs) - np.eye(k)).max() < 1e-9
    assert np.abs(As @ vecs - (Ms @ vecs) * vals).max() < 1e-5 and res[-1].max() < res[0].max() * 1e-3
    # the point of the generalised pencil: the lowest eigenvalue of the box (continuum: 2 in these units, Dirichlet ghosts
    # at -1 and g: the box has g + 1 intervals of h = 1/g, so 2 g^2/(g+1)^2) is closer than the 5-point operator's
    exact = 2.0 * g ** 2 / (g + 1.0) ** 2
    five = drivers.exact_box_eigenvalues(g, "2d", 1)[0]
    assert abs(vals[0] - exact) < 0.05 * abs(five - exact)


@pytest.mark.parametrize("with_mass", [False, True])
def test_rq_line_step_against_the_dense_two_by_two_problem(backend, with_mass):
    """Plan.rq_line_step (mgcmt_rq_line_step): the minimiser of the Rayleigh quotient over span{x, w} — the 2 x 2 pencil of
    MGCMTSolver.py:33-50 solved with scipy's eigh here — x + delta w, its gradient 2 (A x' - rho M x') (:52-54) and the
    recorded Rayleigh quotient; with and without a mass operator, and the direction-free form (rho and g of x alone)."""
    import scipy.linalg
    import scipy.sparse as sp
    from multigridcmt_amd import _lib
    from multigridcmt_amd.operators import potential_well_operator, recognise
    from multigridcmt_amd.plan import get_plan
    g = 32
    op = potential_well_operator(g, 30.0, (8, 24))
    A = op.tocsr()
    m1 = sp.diags([np.full(g - 1, 1 / 6), np.full(g, 2 / 3), np.full(g - 1, 1 / 6)], [-1, 0, 1])
    M = sp.kron(m1, m1).tocsr() if with_mass else sp.eye(g * g, format="csr")
    plan = get_plan(op, 8, nvec=3, mass=recognise(M, "2d") if with_mass else None)
    rng = np.random.RandomState(5)
    x, w = rng.rand(g * g), rng.rand(g * g) - 0.5
    V, F, W = _lib.SLOT_V, _lib.SLOT_F, _lib.SLOT_W
    X, PW, XO, G, WORK = (W, 1), (V, 0), (W, 0), (F, 0), (W, 2)
    plan.upload(0, X[0], X[1], x)
    plan.upload(0, PW[0], PW[1], w)
    plan.rq_line_step(0, X, None, None, G, work=WORK if with_mass else None, record=0)
    rho0 = x @ (A @ x) / (x @ (M @ x))
    assert rel_err(plan.download(0, G[0], G[1]), 2 * (A @ x - rho0 * (M @ x))) < 1e-12
    plan.rq_line_step(0, X, PW, XO, G, work=WORK if with_mass else None, record=1)
    B = np.stack([x, w], axis=1)
    evals, evecs = scipy.linalg.eigh(B.T @ (A @ B), B.T @ (M @ B))
    y = evecs[:, 0]
    xn = x + (y[1] / y[0]) * w
    rho = xn @ (A @ xn) / (xn @ (M @ xn))
    assert abs(rho - evals[0]) < 1e-12 * abs(rho)
    assert rel_err(plan.download(0, XO[0], XO[1]), xn) < 1e-11
    gn = 2 * (A @ xn - rho * (M @ xn))
    assert np.linalg.norm(plan.download(0, G[0], G[1]) - gn) < 1e-10 * np.linalg.norm(2 * (A @ xn))
    assert np.allclose(plan.rq_history(0, 2), [rho0, rho], rtol=1e-12, atol=0)
    assert np.array_equal(plan.download(0, X[0], X[1]), x)            # x itself is only read
    with pytest.raises(_lib.MgcmtError):
        plan.rq_line_step(0, X, PW, X, G)                              # the vectors must be distinct
    if with_mass:
        with pytest.raises(_lib.MgcmtError):
            plan.rq_line_step(0, X, PW, XO, G)                         # a mass operator needs the work vector
