"""Python-3 counterparts of the compute sections of the reference's driver scripts (no plotting).

* ``shift_invert_eigenpairs``  — 1DPotMatrixVcycle.py:42-80 / 2DPotMatrixVcycle.py:54-109: coarse-grid Lanczos
  guesses, interpolation to the fine grid, then ``max_iters`` rounds of  w = vcycle_matrix(0, V, H, shifts=guesses),
  V = normalised columns of w  (shift-and-invert iteration whose linear solves are single V-cycles).
* ``rayleigh_quotient_multigrid`` — RQMin.py:15-50: ``vcycle_rqmg`` sweeps for the lowest pair with Gram-Schmidt
  deflation.

Every V-cycle, interpolation, Gram-Schmidt and operator application runs on the GPU through MGCMTSolver /
MGCMTProcessor; the coarse-grid ``eigsh`` stays on the host as in the reference (16 .. 256 unknowns).
"""
import numpy as np
import scipy.linalg
import scipy.sparse as sparse
import scipy.sparse.linalg as sparsela

from .processor import MGCMTProcessor
from .solver import MGCMTSolver
from .stencil_maker import MGCMTStencilMaker


def exact_box_eigenvalues(gridsize, dimension, count):
    """Eigenvalues of -laplacian(gridsize)/pi^2 (the discrete operator, not the continuum n^2): in 1-D
    (4 g^2/pi^2) sin^2(k pi / (2 (g+1))), k = 1..g; in 2-D all pairwise sums, sorted."""
    k = np.arange(1, gridsize + 1)
    one_d = (4.0 * gridsize ** 2 / np.pi ** 2) * np.sin(k * np.pi / (2.0 * (gridsize + 1))) ** 2
    if dimension == "1d":
        return one_d[:count]
    m = min(gridsize, count + 2)
    return np.sort(np.add.outer(one_d[:m], one_d[:m]).ravel())[:count]


def shift_invert_eigenpairs(dimension="1d", gridsize=2 ** 7, bad_gridsize=2 ** 4, num_eigenvalues=10, max_iters=10,
                            lowest_level=None, tolerance=1e-4, guesses=None, smoother=None, verbose=False):
    """Returns dict(eigenvalues, eigenvectors, guess_eigenvalues, history) for H = -laplacian/pi^2 on a box.

    Defaults are 1DPotMatrixVcycle.py's (128 points, guesses from 16, ten pairs, ten iterations, coarsest V-cycle
    level 16); 2DPotMatrixVcycle.py uses gridsize 64, bad_gridsize 16, max_iters 5, lowest_level 8,
    tolerance = machine epsilon.  ``guesses`` = (values, vectors) skips the Lanczos step (ARPACK's start vector is
    random, so fixtures store its output).
    """
    stencil_maker, solver = MGCMTStencilMaker(), MGCMTSolver()
    if lowest_level is None:
        lowest_level = 2 ** 4 if dimension == "1d" else 2 ** 3
    hamiltonian = (-1 / np.pi ** 2) * stencil_maker.laplacian(gridsize, dimension=dimension)
    n = hamiltonian.shape[0]
    if guesses is None:
        bad_hamiltonian = (-1. / np.pi ** 2) * stencil_maker.laplacian(bad_gridsize, dimension=dimension)
        bad_eigenvalues, bad_eigenvectors = sparsela.eigsh(bad_hamiltonian, k=num_eigenvalues, which="SM", tol=tolerance)
    else:
        bad_eigenvalues, bad_eigenvectors = guesses
    bad_eigenvectors = np.array(bad_eigenvectors)
    vectors = np.zeros((n, num_eigenvalues))
    for j in range(num_eigenvalues):
        vectors[:, j] = solver.interpolate(bad_eigenvectors[:, j], stencil_maker, gridsize, dimension=dimension)
        vectors[:, j] /= np.linalg.norm(vectors[:, j])
    history = np.zeros((max_iters + 1, num_eigenvalues))
    history[0] = [np.dot(vectors[:, j], hamiltonian.dot(vectors[:, j])) for j in range(num_eigenvalues)]
    w0 = np.zeros((n, num_eigenvalues))
    for it in range(1, max_iters + 1):
        w = solver.vcycle_matrix(w0, vectors, hamiltonian, stencil_maker, shifts=bad_eigenvalues, smoother=smoother,
                                 lowest_level=lowest_level, dimension=dimension)
        for j in range(num_eigenvalues):
            vectors[:, j] = w[:, j] / np.linalg.norm(w[:, j])
            history[it, j] = np.dot(vectors[:, j], hamiltonian.dot(vectors[:, j]))
        if verbose:
            print(it, history[it])
    return {"eigenvalues": history[-1].copy(), "eigenvectors": vectors, "guess_eigenvalues": np.array(bad_eigenvalues),
            "history": history}


def nested_iteration_guesses(plan, k, coarse_level, kind, omega, iterations_per_level=1):
    """Eigenpair guesses by nested iteration (full multigrid for the eigenproblem: the reference's report names it as the
    next step — PDF p.17 "FMG", p.51 "Volgende stappen" — its code starts from straight interpolation of coarse Lanczos
    vectors instead, 2DPotMatrixVcycle.py:59-85).  The k lowest eigenpairs of the plan's Galerkin operator on
    ``coarse_level`` are computed densely on the host (a few hundred unknowns, like the reference's coarse ``eigsh``);
    on the way up every level takes the interpolated vectors and improves them by ``iterations_per_level`` rounds of
    the reference's own outer iteration (k-column V-cycle with the coarse eigenvalues as shifts and Gram-Schmidt, then
    normalisation) on THAT level's operator.  Leaves the guesses in slot F of level 0 (normalised) and returns the
    shifts.  Parity: unpinned (no reference counterpart); checked against the straight-interpolation guesses by the
    residuals it produces (tests/test_drivers.py)."""
    from . import _lib
    from .operators import StructuredOperator
    V, F = _lib.SLOT_V, _lib.SLOT_F
    xf = plan.factors(coarse_level, 0) if plan.dim == 2 else None
    yf = plan.factors(coarse_level, 1)
    terms = [((xf[m].copy() if xf is not None else None), yf[m].copy()) for m in range(yf.shape[0])]
    dense = StructuredOperator("2d" if plan.dim == 2 else "1d", plan.g >> coarse_level, terms).tocsr().toarray()
    values, vectors = scipy.linalg.eigh(0.5 * (dense + dense.T))
    shifts = values[:k].copy()
    for j in range(k):
        plan.upload(coarse_level, F, j, np.ascontiguousarray(vectors[:, j]))
    for l in range(coarse_level - 1, -1, -1):
        for j in range(k):
            plan.prolong(l, (F, j), (F, j), accumulate=False)
        plan.normalize(l, F, k)
        if l > 0:
            for _ in range(int(iterations_per_level)):
                plan.set_shifts(shifts)
                plan.vcycle(4, 4, kind, omega=omega, k=k, nu_coarse=4, gram_schmidt=True, level=l, zero_start=True)
                plan.normalize(l, V, k)
                for j in range(k):
                    plan.copy(l, V, j, F, j)
    return shifts


def shift_invert_eigenpairs_resident(dimension="2d", gridsize=2 ** 10, bad_gridsize=2 ** 4, num_eigenvalues=10, max_iters=5,
                                     lowest_level=None, tolerance=1e-4, guesses=None, smoother=None, stats=None,
                                     gram_schmidt="inside", residuals=None, guess_method="interpolate"):
    """The outer loops of 1DPotMatrixVcycle.py:42-80 / 2DPotMatrixVcycle.py:54-109 / 1DPotMGS.py:50-127 with everything
    between the coarse-grid guesses and the final download resident on the GPU: the guesses are interpolated level by
    level inside the plan; every iteration is one k-column V-cycle (one shift per column), a column normalisation, k
    device copies, and ONE batched pass that yields all k Rayleigh quotients and residual norms ||(H - mu_j) v_j||
    (2DPotMatrixVcycle.py:100-105) with a single synchronisation (mgcmt_rayleigh_residual).

    gram_schmidt: where the columns are orthogonalised — the three variants 1DPotMGS.py compares:
      "inside"  at every level on the way up of the cycle (vcycle_matrix, MGCMTSolver.py:434; 1DPotMGS.py:104-124)
      "after"   once per outer iteration, after normalisation and after the Rayleigh quotients are taken (:77-98)
      "none"    never: k independent single-vector cycles (:50-72)
    guess_method: "interpolate" (the reference: coarse eigenvectors interpolated straight to the fine grid) or "fmg"
    (``nested_iteration_guesses``; ``guesses`` is then ignored and bad_gridsize names the coarsest eigen-grid).
    Returns dict(eigenvalues, eigenvectors, guess_eigenvalues, history[, residual_history]); ``residuals`` (a list)
    also receives the residual norms after every iteration; ``stats`` the seconds spent in the iteration loop."""
    import time
    from . import _lib
    from .operators import laplacian_operator
    from .plan import Plan
    if gram_schmidt not in ("inside", "after", "none"):
        raise ValueError("gram_schmidt must be 'inside', 'after' or 'none'")
    stencil_maker, solver = MGCMTStencilMaker(), MGCMTSolver()
    g, k = int(gridsize), int(num_eigenvalues)
    if lowest_level is None:
        lowest_level = 2 ** 4 if dimension == "1d" else 2 ** 3
    lowest_level = min(int(lowest_level), int(bad_gridsize))
    kind, omega = solver._resolve_smoother(smoother)
    if kind is None:
        raise NotImplementedError("the device-resident loop takes MGCMTSolver's own smoothers; use shift_invert_eigenpairs for a callable")
    V, F = _lib.SLOT_V, _lib.SLOT_F
    plan = Plan(laplacian_operator(g, dimension) * (-1 / np.pi ** 2), lowest_level, nvec=k)
    try:
        lb = (g // int(bad_gridsize)).bit_length() - 1          # level of the guess grid
        if guess_method == "fmg":
            bad_eigenvalues = nested_iteration_guesses(plan, k, lb, kind, omega)
        elif guess_method == "interpolate":
            if guesses is None:
                bad_hamiltonian = (-1. / np.pi ** 2) * stencil_maker.laplacian(bad_gridsize, dimension=dimension)
                bad_eigenvalues, bad_eigenvectors = sparsela.eigsh(bad_hamiltonian, k=k, which="SM", tol=tolerance)
            else:
                bad_eigenvalues, bad_eigenvectors = guesses
            bad_eigenvalues, bad_eigenvectors = np.asarray(bad_eigenvalues, dtype=float), np.array(bad_eigenvectors, dtype=float)
            for j in range(k):
                plan.upload(lb, F, j, bad_eigenvectors[:, j])
                for l in range(lb - 1, -1, -1):
                    plan.prolong(l, (F, j), (F, j), accumulate=False)
            plan.normalize(0, F, k)
        else:
            raise ValueError("guess_method must be 'interpolate' or 'fmg'")
        plan.set_shifts(bad_eigenvalues)

        history = np.zeros((max_iters + 1, k))
        residual_history = np.zeros((max_iters + 1, k))
        history[0], residual_history[0] = plan.rayleigh_residual(0, F, k)
        plan.sync()
        start = time.perf_counter()
        for it in range(1, max_iters + 1):
            plan.vcycle(4, 4, kind, omega=omega, k=k, nu_coarse=4, gram_schmidt=(gram_schmidt == "inside"), zero_start=True)
            plan.normalize(0, V, k)
            for j in range(k):
                plan.copy(0, V, j, F, j)
            history[it], residual_history[it] = plan.rayleigh_residual(0, F, k)
            if residuals is not None:
                residuals.append(residual_history[it].copy())
            if gram_schmidt == "after":
                plan.gramschmidt(0, F, k, modified=1)        # processor.gramschmidt(eigenvectors), 1DPotMGS.py:98
        plan.sync()
        if stats is not None:
            stats["loop_seconds"] = time.perf_counter() - start
        vectors = np.stack([plan.download(0, F, j) for j in range(k)], axis=1)
    finally:
        plan.close()
    return {"eigenvalues": history[-1].copy(), "eigenvectors": vectors, "guess_eigenvalues": np.asarray(bad_eigenvalues),
            "history": history, "residual_history": residual_history}


def rayleigh_quotient_multigrid(gridsize=2 ** 6, first_cycles=2, second_cycles=10, seed=0):
    """RQMin.py:15-50 for the two lowest states of -laplacian(gridsize)/pi^2 with M = I: ``first_cycles`` sweeps of
    vcycle_rqmg on a random start, then ``second_cycles`` sweeps on a second random vector with Gram-Schmidt against
    the first after every sweep.  Returns (rho1, rho2, X)."""
    stencil_maker, solver, processor = MGCMTStencilMaker(), MGCMTSolver(), MGCMTProcessor()
    A = (-1 / np.pi ** 2) * stencil_maker.laplacian(gridsize)
    M = sparse.eye(gridsize)
    rng = np.random.RandomState(seed)
    x = rng.random_sample(gridsize)
    rho = 0.0
    for _ in range(first_cycles):
        x, rho = solver.vcycle_rqmg(x, A, M)
    x_matrix = np.zeros((gridsize, 2))
    x_matrix[:, 0] = x
    x_matrix[:, 1] = rng.random_sample(gridsize)
    rho2 = 0.0
    for _ in range(second_cycles):
        x_matrix[:, 1], rho2 = solver.vcycle_rqmg(x_matrix[:, 1], A, M)
        x_matrix = processor.gramschmidt(x_matrix)
    return rho, rho2, x_matrix


def potential_well_eigensolve(gridsize=2 ** 7, depth=50.0, inner=None, cycles=8, method="vcycle", nu=2, lowest=8,
                              smoother="rb", seed=0, history=None, stats=None):
    """BASELINE config 5: the ground state of the 2-D square well  H = -laplacian/pi^2 + V  (V = `depth` outside the
    central square, PotWellSolver.py:150-153 carried to 2-D) by Rayleigh-quotient minimisation on the GPU.

    method="rqmg"    ``cycles`` sweeps of MGCMTSolver.vcycle_rqmg (RQMin.py:25-27 with the 2-D transfers).
    method="vcycle"  Rayleigh-quotient minimisation with a V-cycle preconditioner: every iteration applies one V(nu,nu)
                     cycle of H (from a zero start) to the gradient g = 2 (H x - rho x) and minimises the Rayleigh
                     quotient over span{x, w} — the 2x2 problem of rqmin (MGCMTSolver.py:33-50) with the
                     preconditioned gradient as the search direction — by the two passes of the device-resident rqmin
                     (Plan.rq_line_step: the eight inner products with H w formed in registers, the pencil solved on
                     the device, then x + delta w and its gradient in one pass).  Per iteration: one V-cycle from a zero
                     start (a flag: V is neither cleared nor read) and 48 B per point of vector traffic beside it;
                     nothing comes back to the host inside the loop (the Rayleigh quotients are recorded on the
                     device and fetched once).
    Returns (rho, x); ``history`` (a list) receives rho after every cycle, ``stats`` (a dict) the seconds spent in
    the iteration loop alone (start vector generation and the host transfers excluded)."""
    from . import _lib
    from .operators import identity_operator, potential_well_operator
    from .plan import get_plan
    g = int(gridsize)
    if inner is None:
        inner = (g // 4, 3 * g // 4)
    op = potential_well_operator(g, depth, inner)
    rng = np.random.RandomState(seed)
    x0 = rng.random_sample(g * g)
    if method == "rqmg":
        solver, M = MGCMTSolver(), identity_operator(g, "2d")
        rho = 0.0
        for _ in range(cycles):
            x0, rho = solver.vcycle_rqmg(x0, op, M, nu1=nu, nu2=nu, nmin=lowest)
            if history is not None:
                history.append(rho)
        return rho, x0
    if method != "vcycle":
        raise ValueError("method must be 'rqmg' or 'vcycle'")
    kind, omega = (_lib.GS_MC, 1.0) if smoother == "rb" else (_lib.WJACOBI, 2. / 3.)
    V, F, W = _lib.SLOT_V, _lib.SLOT_F, _lib.SLOT_W
    # the iterate alternates between the two columns of slot W (x + delta w is written beside x); the V-cycle runs on
    # column 0 with shift 0: right-hand side (F, 0) = the gradient, result (V, 0) = the search direction
    X, XALT, PW, G = (W, 1), (W, 0), (V, 0), (F, 0)
    plan = get_plan(op, int(lowest), nvec=2)
    plan.set_shifts([0.0, 0.0])
    plan.upload(0, X[0], X[1], x0)
    plan.scale(0, 1.0 / np.sqrt(plan.dot(0, X, X)), X)
    plan.rq_line_step(0, X, None, None, G)                                            # rho and g = 2 (H x - rho x) of the start vector
    import time
    loop_start = time.perf_counter()
    for it in range(cycles):
        if it and it % _lib.RQ_HISTORY == 0 and history is not None:
            history.extend(plan.rq_history(0, _lib.RQ_HISTORY))
        plan.vcycle(nu, nu, kind, omega=omega, nu_coarse=nu, zero_start=True)        # w = B g  (V is not read: a flag)
        plan.rq_line_step(0, X, PW, XALT, G, record=it % _lib.RQ_HISTORY)             # x <- x + delta w, g <- its gradient
        X, XALT = XALT, X
        if it % 64 == 63:
            # the iterate is never normalised inside the step (nothing in the 2x2 problem depends on its length); keep it
            # of order one over long runs: x and its gradient scale together
            scale = 1.0 / np.sqrt(plan.dot(0, X, X))
            plan.scale(0, scale, X)
            plan.scale(0, scale, G)
    SCRATCH = XALT
    if history is not None and cycles > 0:
        history.extend(plan.rq_history(0, (cycles - 1) % _lib.RQ_HISTORY + 1))
    else:
        plan.sync()
    if stats is not None:
        stats["loop_seconds"] = time.perf_counter() - loop_start
    if cycles > 0:
        # what is returned is MEASURED on the returned vector — <x, H x>/<x, x> — not the running value of the 2x2 problems
        # (which the history holds; the two agree to the accuracy fp64 gives this quotient: eps * |H| / rho, 1e-10 at 8192^2)
        plan.apply(0, X, SCRATCH)
        gram = plan.gram(0, [X, SCRATCH])
        rho = float(gram[0, 1] / gram[0, 0])
        plan.scale(0, 1.0 / np.sqrt(gram[0, 0]), X)                                    # returned with unit length
    else:
        plan.apply(0, X, SCRATCH)
        rho = plan.dot(0, X, SCRATCH)
    return rho, plan.download(0, X[0], X[1])


def block_eigensolve(op, k=4, cycles=12, nu=2, lowest=8, smoother="rb", seed=0, guesses=None, history=None, residuals=None,
                     use_p=True, stats=None, mass=None):
    """SURVEY par. 8(f)4: the k lowest eigenpairs of a structured 2-D operator — of the pencil (A, M) with ``mass`` — by
    BLOCKED Rayleigh-Ritz with a V-cycle preconditioner: the reference's Rayleigh-quotient routines (rqmin's 2 x 2 problem
    over span{x, p} with its generalised form R y = lambda RM y, MGCMTSolver.py:33-50; the dead vcycle_rqmg2, :59-94, which
    carries M alongside A, :78-79) carried to a block of k vectors with LOBPCG-style updates.  Not in the reference: parity
    unpinned; checked against exact eigenvalues and scipy's eigsh.

    Per iteration, all vectors resident in HBM:  R = A X - M X diag(rho);  W = one V(nu, nu) cycle of A per column of R
    from a zero start (k columns batched);  A W, M W;  Rayleigh-Ritz over S = [X, W, P] (P: the previous update directions;
    use_p=False gives blocked preconditioned steepest descent): the 3k x 3k pencil (S^T A S, S^T M S) from one-pass block
    Gram products, its k lowest pairs on the host, then X, P and their images under A and M updated by tall-skinny
    products.  The host sees Gram matrices only.  k <= 4.  ``mass``: a StructuredOperator (None: the identity — nothing of
    M is then stored or applied).

    Returns (eigenvalues, eigenvectors[n, k]) with X^T M X = I; ``history`` receives the Ritz values after every iteration,
    ``residuals`` the norms ||A x_j - rho_j M x_j||."""
    from . import _lib
    from ._lib import OP_M
    from .plan import get_plan
    k = int(k)
    if not 1 <= k <= 4:
        raise ValueError("block_eigensolve handles 1..4 eigenpairs at a time")
    kind, omega = (_lib.GS_MC, 1.0) if smoother == "rb" else (_lib.WJACOBI, 2. / 3.)
    V, F, W = _lib.SLOT_V, _lib.SLOT_F, _lib.SLOT_W
    gen = mass is not None
    plan = get_plan(op, int(lowest), nvec=(5 if gen else 3) * k, mass=mass)
    plan.set_shifts(np.zeros(plan.nvec))
    n = plan.size(0)
    # what must survive a cycle lives in slot W and in the columns of slot F the cycle does not use (a cycle may exchange
    # the storage of slots V and T, and uses T as its scratch)
    X = [(W, j) for j in range(k)]
    AX = [(W, k + j) for j in range(k)]
    P = [(W, 2 * k + j) for j in range(k)]
    R = [(F, j) for j in range(k)]           # the cycle's right-hand side
    AP = [(F, k + j) for j in range(k)]
    AW = [(F, 2 * k + j) for j in range(k)]
    Wd = [(V, j) for j in range(k)]          # the cycle's output
    # images under M (M = I: the vectors themselves)
    MX = [(W, 3 * k + j) for j in range(k)] if gen else X
    MP = [(W, 4 * k + j) for j in range(k)] if gen else P
    MW = [(F, 3 * k + j) for j in range(k)] if gen else Wd
    rng = np.random.RandomState(seed)
    for j in range(k):
        plan.upload(0, W, j, rng.random_sample(n) if guesses is None else np.asarray(guesses)[:, j])

    def ritz(S, AS, MS, take):
        """k lowest Ritz pairs of the pencil (S^T A S, S^T M S); blocks of at most four columns per Gram product"""
        m = len(S)
        G, H = np.zeros((m, m)), np.zeros((m, m))
        for c in range(0, m, 4):
            G[:, c:c + 4] = plan.block_gram(0, S, MS[c:c + 4])
            H[:, c:c + 4] = plan.block_gram(0, S, AS[c:c + 4])
        G, H = 0.5 * (G + G.T), 0.5 * (H + H.T)
        d = 1.0 / np.sqrt(np.diag(G))                   # (scaling only: the pencil's eigenvectors are rescaled back)
        evals, evecs = scipy.linalg.eigh(H * np.outer(d, d), G * np.outer(d, d), subset_by_index=[0, take - 1])
        return evals, evecs * d[:, None]

    import time
    for j in range(k):
        plan.apply(0, X[j], AX[j])
        if gen:
            plan.apply(0, X[j], MX[j], op=OP_M)
    rho, C = ritz(X, AX, MX, k)
    plan.block_combine(0, X, X, C)
    plan.block_combine(0, AX, AX, C)
    if gen:
        plan.block_combine(0, MX, MX, C)
    have_p = False
    loop_start = time.perf_counter()
    for _ in range(int(cycles)):
        for j in range(k):
            plan.lincomb(0, [(1.0, AX[j]), (-float(rho[j]), MX[j])], R[j])
        if residuals is not None:
            residuals.append(np.sqrt(np.diag(plan.block_gram(0, R, R))))
        plan.vcycle(nu, nu, kind, omega=omega, k=k, nu_coarse=nu, zero_start=True)
        for j in range(k):
            plan.apply(0, Wd[j], AW[j])
            if gen:
                plan.apply(0, Wd[j], MW[j], op=OP_M)
        if have_p:
            S, AS, MS = X + Wd + P, AX + AW + AP, MX + MW + MP
        else:
            S, AS, MS = X + Wd, AX + AW, MX + MW
        try:
            rho, C = ritz(S, AS, MS, k)
        except (scipy.linalg.LinAlgError, ValueError):   # the directions have become dependent: restart without P
            S, AS, MS = X + Wd, AX + AW, MX + MW
            rho, C = ritz(S, AS, MS, k)
        # P' = [W, P] C_wp, X' = X C_x + P'  (and the same for their images under A and M): P is overwritten first, X uses the new P
        plan.block_combine(0, S[k:], P, C[k:])
        plan.block_combine(0, AS[k:], AP, C[k:])
        plan.block_combine(0, X + P, X, np.vstack([C[:k], np.eye(k)]))
        plan.block_combine(0, AX + AP, AX, np.vstack([C[:k], np.eye(k)]))
        if gen:
            plan.block_combine(0, MS[k:], MP, C[k:])
            plan.block_combine(0, MX + MP, MX, np.vstack([C[:k], np.eye(k)]))
        have_p = bool(use_p)
        if history is not None:
            history.append(np.array(rho))
    plan.sync()
    if stats is not None:
        stats["loop_seconds"] = time.perf_counter() - loop_start
    vecs = np.stack([np.asarray(plan.download(0, W, j)) for j in range(k)], axis=1)
    return np.array(rho), vecs
