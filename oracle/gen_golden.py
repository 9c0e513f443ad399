"""Writes tests/golden/*.npz by RUNNING THE REFERENCE ITSELF (container-only; see ref_loader.py).

TEST INFRASTRUCTURE — NOT PRODUCT CODE.

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python -m oracle.gen_golden

Each fixture holds inputs and the reference's outputs only (arrays and scalars) — no reference
source text.  Cases follow the reference's own tests (UnitTests/*.py), the loader-sanity values of
SURVEY.md §8(c), and re-enactments of the compute sections of 1DPotMatrixVcycle.py:42-80,
2DPotMatrixVcycle.py:54-109 and RQMin.py:15-50 at sizes the reference finishes in seconds.  ARPACK
guesses and random inputs are stored as fixture INPUTS (eigsh start vectors are unseeded).
"""
import os
import sys
import warnings

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as sla

from oracle import ref_loader
from oracle.sparse_ref import RefSolver

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _c(x):
    return np.array(x, dtype=float).copy()


def main():
    warnings.simplefilter("ignore")
    Solver, StencilMaker, Processor = ref_loader.load_reference()
    solver, sm, proc = Solver(), StencilMaker(), Processor()
    os.makedirs(OUT, exist_ok=True)
    devnull = open(os.devnull, "w")

    only = set(sys.argv[1:])               # optional: names of the fixtures to (re)write; default all

    def save(name, **kw):
        if only and name not in only:
            return
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **kw)
        print("wrote", name, {k: np.shape(v) for k, v in kw.items()}, file=sys.stderr)

    # ---- UnitTests/{wjacobi,gseidel,sor,vcycle,twogrid}Test.py: A = laplacian(16), f = 0, x0 = 1 ----
    n = 16
    A = sm.laplacian(n)
    kat = {}
    x = np.ones((n, 1))
    for _ in range(5):
        x = solver.wjacobi(x, np.zeros((n, 1)), A, nu=4)
    kat["wjacobi_x"] = _c(x).ravel()
    x = np.ones((n, 1))
    for _ in range(5):
        x = solver.gseidel(x, np.zeros((n, 1)), A)
    kat["gseidel_x"] = _c(x).ravel()
    x = np.ones((n, 1))
    for _ in range(5):
        x = solver.sor(x, np.zeros((n, 1)), A, nu=4, omega=2. / 3.)
    kat["sor_x"] = _c(x).ravel()
    kat["vcycle_x"] = _c(solver.vcycle(np.ones((n, 1)), np.zeros((n, 1)), A, sm, nu1=4, nu2=4)).ravel()
    kat["twogrid_x"] = _c(solver.twogrid(np.ones((n, 1)), np.zeros((n, 1)), A, sm, nu1=4, nu2=4)).ravel()
    # expected norms carried as comments in the reference's tests (UnitTests/*Test.py:25)
    kat["expected_norms"] = np.array([2.94959, 1.88358, 2.63327, 0.17756, 0.04979])
    save("kat_unit_tests", **kat)

    # ---- UnitTests/vcycle_matrixTest.py:21-39 (n = 4, 3 right-hand sides) ----
    n, k = 4, 3
    lap = sm.laplacian(n)
    F = np.zeros((n, k))
    for i in range(k):
        F[:, i] = np.ones(n) * i
    Xv = np.ones((n, k)) * 4
    Xt = np.ones((n, k)) * 4
    for i in range(k):
        Xv[:, i] = solver.vcycle(_c(Xv[:, i]), _c(F[:, i]), lap, sm)
        Xt[:, i] = solver.twogrid(_c(Xt[:, i]), _c(F[:, i]), lap, sm)
    Xm = solver.vcycle_matrix(np.ones((n, k)) * 4, F.copy(), lap, sm, shifts=np.zeros(k))
    save("kat_vcycle_matrix_n4", F=F, x_vcycle=Xv, x_twogrid=Xt, x_vcycle_matrix=Xm,
         quad_vcycle=np.array([Xv[:, j] @ (lap @ Xv[:, j]) for j in range(k)]),
         quad_comment=np.array([-0.382301639189, -0.0257586075778, -0.840838733687]))

    # ---- operators (UnitTests/operatorTest.py, 2DInterGridTest.py) ----
    ops = {}
    ops["R_16_8"] = sm.restriction(16, 8).toarray()
    ops["P_8_16"] = sm.interpolation(8, 16).toarray()
    ops["RAP_16"] = (sm.restriction(16, 8) * sm.laplacian(16) * sm.interpolation(8, 16)).toarray()
    ops["P_4_16"] = sm.interpolation(4, 16).toarray()
    ops["R_16_4"] = sm.restriction(16, 4).toarray()
    ops["P2d_4_8"] = sm.interpolation(4, 8, dimension="2d").toarray()
    ops["R2d_8_4"] = sm.restriction(8, 4, dimension="2d").toarray()
    ops["R2d_16_4"] = sm.restriction(16, 4, dimension="2d").toarray()
    ops["P2d_4_16"] = sm.interpolation(4, 16, dimension="2d").toarray()
    ops["L1d_8"] = sm.laplacian(8).toarray()
    ops["L2d_4"] = sm.laplacian(4, dimension="2d").toarray()
    R2, P2, L2 = sm.restriction(8, 4, dimension="2d"), sm.interpolation(4, 8, dimension="2d"), sm.laplacian(8, dimension="2d")
    ops["RAP2d_8"] = (R2 * L2 * P2).toarray()
    save("operators", **ops)

    # ---- smoothers on random data, 1-D and 2-D, with scale and shift ----
    rng = np.random.RandomState(7)
    sm_cases = {}
    for tag, dim, g in (("1d64", "1d", 64), ("2d16", "2d", 16)):
        H = (-1 / np.pi ** 2) * sm.laplacian(g, dimension=dim)
        N = H.shape[0]
        Ash = H - sp.eye(N) * 1.7
        v0, f = rng.rand(N), rng.rand(N)
        sm_cases[tag + "_v0"], sm_cases[tag + "_f"] = v0, f
        sm_cases[tag + "_wjacobi"] = _c(solver.wjacobi(_c(v0), _c(f), Ash, nu=3, omega=0.8)).ravel()
        sm_cases[tag + "_gseidel"] = _c(solver.gseidel(_c(v0).reshape(N, 1), _c(f).reshape(N, 1), Ash, nu=3)).ravel()
        sm_cases[tag + "_sor"] = _c(solver.sor(_c(v0).reshape(N, 1), _c(f).reshape(N, 1), Ash, nu=3, omega=1.5)).ravel()
    sm_cases["shift"] = np.array(1.7)
    save("smoothers_random", **sm_cases)

    # ---- vcycle / twogrid: 1-D ----
    vc = {}
    A1024 = sm.laplacian(1024)
    vc["cfg1_twogrid_gs"] = _c(solver.twogrid(np.ones((1024, 1)), np.zeros((1024, 1)), A1024, sm, nu1=4, nu2=4,
                                               smoother=solver.gseidel)).ravel()
    vc["cfg1_vcycle512_gs"] = _c(solver.vcycle(np.ones((1024, 1)), np.zeros((1024, 1)), A1024, sm, nu1=4, nu2=4,
                                                smoother=solver.gseidel, lowest_level=512)).ravel()
    vc["cfg1_norm_survey"] = np.array(0.023328643431521426)
    vc["full1024_gs"] = _c(solver.vcycle(np.ones((1024, 1)), np.zeros((1024, 1)), A1024, sm, nu1=4, nu2=4,
                                          smoother=solver.gseidel)).ravel()
    vc["full1024_norm_survey"] = np.array(0.500560418530017)
    rng = np.random.RandomState(1)
    f1 = rng.rand(1024)
    vc["cfg1b_f"] = f1
    vc["cfg1b_twogrid_gs"] = _c(solver.twogrid(np.zeros(1024), _c(f1), A1024, sm, nu1=4, nu2=4,
                                                smoother=solver.gseidel)).ravel()
    H128 = (-1 / np.pi ** 2) * sm.laplacian(128)
    f2 = rng.rand(128)
    vc["h128_f"] = f2
    for name, smo in (("wj", solver.wjacobi), ("gs", solver.gseidel), ("sor", lambda v, f, A, nu=4: solver.sor(v, f, A, nu=nu, omega=1.3))):
        vc["h128_vcycle_%s_shift0.9_low8" % name] = _c(solver.vcycle(
            np.zeros((128, 1)), _c(f2).reshape(128, 1), H128, sm, nu1=2, nu2=3, smoother=smo, shift=0.9,
            lowest_level=8)).ravel()
    save("vcycle_1d", **vc)

    # ---- vcycle: 2-D (16^2 and 32^2) ----
    vc2 = {}
    rng = np.random.RandomState(0)
    for g in (16, 32):
        H = (-1 / np.pi ** 2) * sm.laplacian(g, dimension="2d")
        f = rng.rand(g * g)
        vc2["g%d_f" % g] = f
        for name, smo in (("wj", solver.wjacobi), ("gs", solver.gseidel)):
            vc2["g%d_%s_shift1.9_low8" % (g, name)] = _c(solver.vcycle(
                np.zeros((g * g, 1)), _c(f).reshape(-1, 1), H, sm, shift=1.9, smoother=smo, dimension="2d",
                lowest_level=8)).ravel()
        vc2["g%d_wj_v22_shift0_low2" % g] = _c(solver.vcycle(
            np.zeros((g * g, 1)), _c(f).reshape(-1, 1), H, sm, nu1=2, nu2=2, dimension="2d")).ravel()
    vc2["g16_norm_survey"] = np.array(37.530067463013886)
    L16 = sm.laplacian(16, dimension="2d")
    vc2["lap16_sor1.2_low4"] = _c(solver.vcycle(
        np.zeros((256, 1)), _c(vc2["g16_f"]).reshape(-1, 1), L16, sm, nu1=3, nu2=1,
        smoother=lambda v, f, A, nu=4: solver.sor(v, f, A, nu=nu, omega=1.2), dimension="2d", lowest_level=4)).ravel()
    save("vcycle_2d", **vc2)

    # ---- the reference's vcycle with the multicolour smoother injected through smoother= ----
    oracle = RefSolver()
    mc = {}
    rng = np.random.RandomState(3)
    for dim, g in (("1d", 128), ("2d", 16), ("2d", 32)):
        H = (-1 / np.pi ** 2) * sm.laplacian(g, dimension=dim)
        N = H.shape[0]
        f = rng.rand(N)
        smo = (lambda d: (lambda v, f, A, nu=4: oracle.gseidel_mc(v, f, A, nu=nu, dimension=d).reshape(-1, 1)))(dim)
        tag = "%s_g%d" % (dim, g)
        mc[tag + "_f"] = f
        mc[tag + "_v22_shift0.5_low4"] = _c(solver.vcycle(
            np.zeros((N, 1)), _c(f).reshape(-1, 1), H, sm, nu1=2, nu2=2, smoother=smo, shift=0.5,
            dimension=dim, lowest_level=4)).ravel()
    save("vcycle_multicolour_injected", **mc)

    # ---- vcycle_matrix, 1-D and 2-D, with per-column shifts ----
    vm = {}
    rng = np.random.RandomState(5)
    H = (-1 / np.pi ** 2) * sm.laplacian(64)
    Fm = rng.rand(64, 3)
    shifts = np.array([0.9, 3.8, 8.5])
    vm["h64_F"], vm["h64_shifts"] = Fm, shifts
    vm["h64_wj_low8"] = solver.vcycle_matrix(np.zeros((64, 3)), Fm.copy(), H, sm, shifts=shifts, lowest_level=8)
    # gseidel does not reshape its inputs (MGCMTSolver.py:210-227): handed the 1-D column views that
    # vcycle_matrix passes (:416) its `v + result` broadcasts (n,)+(n,1) to an n x n array, which is not
    # a smoother at all.  No reference caller does that; the meaningful call reshapes the columns:
    gs_cols = lambda v, f, A, nu=4: solver.gseidel(np.array(v).reshape(-1, 1), np.array(f).reshape(-1, 1), A, nu=nu)
    vm["h64_gs_low8"] = solver.vcycle_matrix(np.zeros((64, 3)), Fm.copy(), H, sm, shifts=shifts, lowest_level=8,
                                              smoother=gs_cols)
    H2 = (-1 / np.pi ** 2) * sm.laplacian(16, dimension="2d")
    Fm2 = rng.rand(256, 3)
    shifts2 = np.array([1.9, 4.7, 4.8])
    vm["h2d16_F"], vm["h2d16_shifts"] = Fm2, shifts2
    vm["h2d16_wj_low4"] = solver.vcycle_matrix(np.zeros((256, 3)), Fm2.copy(), H2, sm, shifts=shifts2,
                                                lowest_level=4, dimension="2d")
    save("vcycle_matrix", **vm)

    # ---- Gram-Schmidt (UnitTests/GramSchmidt.py:10-67, 73-129) ----
    eps = np.finfo(float).eps
    G1 = np.column_stack((np.array([1, eps, eps]), np.array([1, eps, 0]), np.array([1, 0, eps])))
    G2 = np.column_stack((np.array([1., 2, 5]), np.array([1., 1, 1]), np.array([1., 0, 3])))
    rng = np.random.RandomState(11)
    G3 = rng.rand(200, 6)
    save("gramschmidt", G1=G1, G1_cgs=proc.gramschmidt(G1, modified=0), G1_mgs=proc.gramschmidt(G1, modified=1),
         G2=G2, G2_cgs=proc.gramschmidt(G2, modified=0), G2_mgs=proc.gramschmidt(G2, modified=1),
         G3=G3, G3_cgs=proc.gramschmidt(G3, modified=0), G3_mgs=proc.gramschmidt(G3, modified=1),
         G3_normalize=proc.normalize(G3), G3_gram=proc.orthogonality_check(G3))

    # ---- rqmin / vcycle_rqmg / vcycle_rqmg2 (RQMin.py:15-50) ----
    rq = {}
    A64 = (-1 / np.pi ** 2) * sm.laplacian(64)
    M64 = sp.eye(64)
    x0 = np.random.RandomState(0).rand(64)
    rq["x0"] = x0
    x, rho = solver.rqmin(A64, _c(x0), M64, nu=4)
    rq["rqmin_x"], rq["rqmin_rho"] = np.real(x), np.real(rho)
    rq["rqmin_rho_survey"] = np.array(4.304377712544061)
    x = _c(x0)
    rhos = []
    for _ in range(2):
        x, rho = solver.vcycle_rqmg(x, A64, M64)
        rhos.append(np.real(rho))
    rq["rqmg_x"], rq["rqmg_rhos"] = np.real(x), np.array(rhos)
    rq["rqmg_rho_survey"] = np.array(0.9706044628745033)
    # the two-grid form of the same algorithm (what the dead twogridrqmin, :127-178, is the skeleton of): one coarsening
    x, rho = solver.vcycle_rqmg(_c(x0), A64, M64, nu1=4, nu2=4, nmin=32)
    rq["twogrid_x"], rq["twogrid_rho"] = np.real(x), np.real(rho)
    A32 = (-1 / np.pi ** 2) * sm.laplacian(32)
    X0 = np.random.RandomState(0).rand(32, 2)
    rq["X0"] = X0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        X = solver.vcycle_rqmg2(X0.copy(), A32, sp.eye(32), nmin=4)
    rq["rqmg2_nmin4_X"] = X
    rq["rqmg2_nmin4_rq"] = np.array([X[:, i] @ (A32 @ X[:, i]) / (X[:, i] @ X[:, i]) for i in range(2)])
    rq["rqmg2_nmin4_rq_survey"] = np.array([0.963017800746613, 1.040538684849706])
    save("rqmin", **rq)

    # ---- the same family with a NON-IDENTITY mass operator (the reference coarsens M alongside A: MGCMTSolver.py:78-79,
    # 110-111): M = the linear finite-element mass matrix tridiag(1, 4, 1) / 6 ----
    def fem_mass(n):
        return sp.diags([np.full(n - 1, 1.0 / 6.0), np.full(n, 4.0 / 6.0), np.full(n - 1, 1.0 / 6.0)], [-1, 0, 1], format="csr")

    rm = {"x0": x0, "X0": X0}
    Mt64, Mt32 = fem_mass(64), fem_mass(32)
    x, rho = solver.rqmin(A64, _c(x0), Mt64, nu=4)
    rm["rqmin_x"], rm["rqmin_rho"] = np.real(x), np.real(rho)
    x = _c(x0)
    rhos = []
    for _ in range(2):
        x, rho = solver.vcycle_rqmg(x, A64, Mt64)
        rhos.append(np.real(rho))
    rm["rqmg_x"], rm["rqmg_rhos"] = np.real(x), np.array(rhos)
    x, rho = solver.vcycle_rqmg(_c(x0), A64, Mt64, nu1=4, nu2=4, nmin=32)
    rm["twogrid_x"], rm["twogrid_rho"] = np.real(x), np.real(rho)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        X = solver.vcycle_rqmg2(X0.copy(), A32, Mt32, nmin=4)
    rm["rqmg2_nmin4_X"] = X
    rm["rqmg2_nmin4_rq"] = np.array([X[:, i] @ (A32 @ X[:, i]) / (X[:, i] @ (Mt32 @ X[:, i])) for i in range(2)])
    save("rqmin_mass", **rm)

    # ---- driver re-enactments ----
    # 1DPotMatrixVcycle.py:14-80 at the script's own sizes (n = 128, guess 16, 10 pairs, 10 iterations)
    g, bad, k, iters = 128, 16, 10, 10
    H = (-1 / np.pi ** 2) * sm.laplacian(g)
    Hb = (-1 / np.pi ** 2) * sm.laplacian(bad)
    bad_vals, bad_vecs = sla.eigsh(Hb, k=k, which="SM", tol=1e-4)
    P = sm.interpolation(bad, g)
    V = np.zeros((g, k))
    for j in range(k):
        V[:, j] = P * bad_vecs[:, j]
        V[:, j] /= np.linalg.norm(V[:, j])
    V0 = V.copy()
    hist = np.zeros((iters, k))
    for it in range(iters):
        w = solver.vcycle_matrix(np.zeros((g, k)), V.copy(), H, sm, shifts=bad_vals, lowest_level=2 ** 4)
        for j in range(k):
            V[:, j] = w[:, j] / np.linalg.norm(w[:, j])
            hist[it, j] = V[:, j] @ (H @ V[:, j])
    save("driver_1dpot_matrix_vcycle", bad_vals=bad_vals, bad_vecs=bad_vecs, V0=V0, V_final=V, rq_history=hist)

    # 2DPotMatrixVcycle.py:15-109 reduced to 32^2 / guess 8^2 / 4 pairs / 3 iterations / lowest 4
    g, bad, k, iters = 32, 8, 4, 3
    H = (-1 / np.pi ** 2) * sm.laplacian(g, dimension="2d")
    Hb = (-1. / np.pi ** 2) * sm.laplacian(bad, dimension="2d")
    bad_vals, bad_vecs = sla.eigsh(Hb, which="SM", tol=np.finfo(float).eps, k=k)
    P = sm.interpolation(bad, g, dimension="2d")
    V = np.zeros((g * g, k))
    for j in range(k):
        V[:, j] = P * bad_vecs[:, j]
        V[:, j] /= np.linalg.norm(V[:, j])
    V0 = V.copy()
    hist = np.zeros((iters, k))
    res = np.zeros((iters, k))
    for it in range(iters):
        w = solver.vcycle_matrix(np.zeros((g * g, k)), V.copy(), H, sm, shifts=bad_vals, dimension="2d",
                                 lowest_level=4)
        for j in range(k):
            V[:, j] = w[:, j] / np.linalg.norm(w[:, j])
            res[it, j] = np.linalg.norm((H - sp.eye(g * g) * bad_vals[j]).dot(V[:, j]))
            hist[it, j] = V[:, j] @ (H @ V[:, j])
    save("driver_2dpot_matrix_vcycle", bad_vals=bad_vals, bad_vecs=bad_vecs, V0=V0, V_final=V,
         rq_history=hist, residual_history=res)
    # 1DPotMGS.py:14-127 at the script's own sizes (n = 256, guesses from 8, 6 pairs, 10 iterations, lowest level 8):
    # the three Gram-Schmidt placements it compares — none (:50-72), after every outer iteration (:77-98), inside the
    # cycle (vcycle_matrix, :104-124) — from the same ARPACK guesses
    g, bad, k, iters, low = 2 ** 8, 2 ** 3, 6, 10, 2 ** 3
    H = (-1 / np.pi ** 2) * sm.laplacian(g)
    Hb = (-1 / np.pi ** 2) * sm.laplacian(bad)
    bad_vals, bad_vecs = sla.eigsh(Hb, k=k, which="SM", tol=1e-4)
    bad_vecs = np.array(bad_vecs)
    P = sm.interpolation(bad, g)
    mgs = {"bad_vals": bad_vals, "bad_vecs": bad_vecs}
    for placement in ("none", "after", "inside"):
        V = np.zeros((g, k))
        hist = np.zeros((iters + 1, k))
        for j in range(k):
            V[:, j] = P * bad_vecs[:, j]
            V[:, j] /= np.linalg.norm(V[:, j])
            hist[0, j] = np.dot(V[:, j].conj().T, H.dot(V[:, j]))
        for it in range(1, iters + 1):
            if placement == "inside":
                w = solver.vcycle_matrix(np.zeros((g, k)), V.copy(), H, sm, shifts=bad_vals, lowest_level=low)
                for j in range(k):
                    V[:, j] = w[:, j] / np.linalg.norm(w[:, j])
                    hist[it, j] = np.dot(V[:, j].conj().T, H.dot(V[:, j]))
            else:
                for j in range(k):
                    w = solver.vcycle(np.zeros((g, 1)), V[:, j].copy(), H, sm, shift=bad_vals[j], lowest_level=low)
                    V[:, j] = w / np.linalg.norm(w)
                    hist[it, j] = np.dot(V[:, j].conj().T, H.dot(V[:, j]))
                if placement == "after":
                    V = proc.gramschmidt(V)
        mgs["rq_history_" + placement] = hist
        mgs["V_final_" + placement] = V.copy()
    save("driver_1dpot_mgs", **mgs)

    # ---- the reference's vcycle on its own k.p Hamiltonian (ThesisProblem.py:26-40,62-101) ----------
    # 4-band GaAs well confined along z, complex 4n x 4n block matrix cycled as ONE 1-D grid of length 4n, guesses
    # from 32 points per band, lowest_level = 2**5, smoother = solver.gseidel — at n = 64 and off the gamma point
    # (k = 0.5), where the off-diagonal blocks are non-zero and imaginary
    PotWellSolver, Compound, GaAsValues, PotentialWell = ref_loader.load_kp_model()
    kp = {}
    pws = PotWellSolver(Compound(GaAsValues), PotentialWell("z"), 4)
    for tag, npts, kpoint in (("n64_k0.5", 64, 0.5), ("n128_k0", 128, 0.0)):
        pws.setGridPoints(npts)
        pws.setXRange(-1, 1) if hasattr(pws, "setXRange") else None
        pws.setDense(0)
        Hkp = sp.csr_matrix(pws.makeMatrix(kpoint))
        Hkp.sort_indices()
        N = Hkp.shape[0]
        pws.setGridPoints(32)
        bad = sp.csr_matrix(pws.makeMatrix(kpoint))
        guess = np.sort(sla.eigsh(bad, k=3, which="SM", tol=1e-12)[0])
        rng = np.random.RandomState(len(tag))
        fk = rng.rand(N) + 1j * rng.rand(N)
        kp[tag + "_data"], kp[tag + "_indices"], kp[tag + "_indptr"] = Hkp.data, Hkp.indices, Hkp.indptr
        kp[tag + "_f"], kp[tag + "_shift"] = fk, np.array(guess[0])
        for name, smo in (("gs", solver.gseidel), ("wj", solver.wjacobi), ("sor", lambda v, f, A, nu=4: solver.sor(v, f, A, nu=nu, omega=1.2))):
            kp["%s_vcycle_%s" % (tag, name)] = np.array(solver.vcycle(
                np.zeros((N, 1), dtype=complex), fk.copy().reshape(N, 1), Hkp, sm, shift=guess[0], lowest_level=2 ** 5,
                smoother=smo)).ravel()
        kp[tag + "_gseidel3"] = np.array(solver.gseidel(fk.copy().reshape(N, 1) * 0.5, fk.copy().reshape(N, 1), Hkp, nu=3)).ravel()
        kp[tag + "_wjacobi3"] = np.array(solver.wjacobi(fk.copy().reshape(N, 1) * 0.5, fk.copy().reshape(N, 1), Hkp, nu=3)).ravel()
        R1, P1 = sm.restriction(N, N // 2), sm.interpolation(N // 2, N)
        kp[tag + "_rap_dense"] = (R1 * Hkp * P1).toarray()
    save("kp_well", **kp)

    # ---- BASELINE config 5 at a size the reference can run: square-well Hamiltonian on a 2-D grid -------------------
    # H = -laplacian/pi^2 + diag(V), V = depth outside the square [lo,hi)^2 (the potential of PotWellSolver.py:150-153
    # carried to two dimensions), through the reference's own vcycle (2-D transfers) and rqmin (RQMin.py:18-27)
    pw = {}
    g, depth, lo, hi = 32, 30.0, 8, 24
    chi = np.zeros(g)
    chi[lo:hi] = 1.0
    V = depth * (1.0 - np.outer(chi, chi)).reshape(-1)
    Hw = ((-1 / np.pi ** 2) * sm.laplacian(g, dimension="2d") + sp.diags(V)).tocsr()
    rng = np.random.RandomState(11)
    f, x0 = rng.rand(g * g), rng.rand(g * g)
    pw["g"], pw["depth"], pw["inner"], pw["f"], pw["x0"] = np.array(g), np.array(depth), np.array([lo, hi]), f, x0
    for name, smo in (("wj", solver.wjacobi), ("gs", solver.gseidel)):
        pw["vcycle_%s_v22_shift0.7_low8" % name] = _c(solver.vcycle(
            _c(x0).reshape(-1, 1), _c(f).reshape(-1, 1), Hw, sm, nu1=2, nu2=2, shift=0.7, smoother=smo, dimension="2d",
            lowest_level=8)).ravel()
    x, rho = solver.rqmin(Hw, _c(x0), sp.eye(g * g), nu=6)
    pw["rqmin_x"], pw["rqmin_rho"] = np.real(x), np.real(rho)
    save("potential_well_2d", **pw)
    devnull.close()


if __name__ == "__main__":
    main()
