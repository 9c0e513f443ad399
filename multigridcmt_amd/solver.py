"""MGCMTSolver — the reference's solver class (MGCMTSolver.py:8-436) with every numerical step on
the MI355X through libmgcmt_hip.so.

Signatures (positional order, keyword names, defaults), return shapes and the reference's quirks are
kept; trailing keyword-only additions are marked "addition".  Operators arrive as ``scipy.sparse``
matrices exactly as in the reference's drivers and are mapped to the matrix-free representation by
``operators.recognise``; operators it cannot map raise (there is no CPU fallback).
"""
import functools
import math

import numpy as np
import scipy.linalg

from . import _lib
from ._lib import GS_LEX, GS_MC, OP_A, OP_M, SLOT_F, SLOT_T, SLOT_V, SLOT_W, SOR_LEX, WJACOBI
from .operators import StructuredOperator, UnrecognisedOperator, laplacian_operator, recognise, tag_structured
from .plan import get_plan
from .processor import MGCMTProcessor
from .stencil_maker import MGCMTStencilMaker


def _is_pow2(x):
    x = int(x)
    return x > 0 and (x & (x - 1)) == 0


def _all_zero(a):
    """True when every entry of `a` is zero.  A handful of samples settles the usual non-zero start vector at no cost
    (a full scan of a 0.5 GB iterate costs more than its transfer); only when they are all zero is the array scanned."""
    flat = a.ravel(order="K")                  # a view of contiguous arrays, whatever their order
    n = flat.size
    if n == 0:
        return True
    step = max(1, n // 61)
    if flat[::step].any() or flat[-1] != 0:
        return False
    return not flat.any()


def _force_column(a, n):
    """The reference sets ``.shape = (n, 1)`` on the CALLER's arrays (MGCMTSolver.py:187-191,297-300)."""
    if isinstance(a, np.ndarray) and a.shape != (n, 1):
        try:
            a.shape = (n, 1)
        except AttributeError:
            pass
    return a


class MGCMTSolver:
    """
    Same constructor as the reference (MGCMTSolver.py:13-15): owns a stencil maker and a processor.
    """

    def __init__(self):
        self.stencil_maker = MGCMTStencilMaker()
        self.processor = MGCMTProcessor()

    # ------------------------------------------------------------------------------------------
    # smoothers
    # ------------------------------------------------------------------------------------------
    def _smooth(self, v0, f, A, kind, nu, omega, dimension=None):
        n = len(v0)
        try:
            op = recognise(A, dimension)
        except UnrecognisedOperator:
            if dimension == "2d":
                raise
            return self._smooth_general(v0, f, A, kind, nu, omega)
        plan = get_plan(op, op.g, nvec=1)
        plan.set_shifts([0.0])
        plan.upload(0, SLOT_V, 0, np.asarray(v0, dtype=np.float64).reshape(-1))
        plan.upload(0, SLOT_F, 0, np.asarray(f, dtype=np.float64).reshape(-1))
        plan.smooth(0, kind, int(nu), omega=float(omega), k=1)
        return plan.download(0, SLOT_V, 0).reshape(n, 1)

    # -- general sparse / complex operators: the k.p Hamiltonians of ThesisProblem.py:38-40,80,101 ----
    def _smooth_general(self, v0, f, A, kind, nu, omega):
        from .general import get_csr_plan, real_if_real
        n = len(v0)
        plan = get_csr_plan(A, n)                       # one level: smoothing only
        plan.upload(0, SLOT_V, v0)
        plan.upload(0, SLOT_F, f)
        plan.smooth(0, kind, int(nu), omega=float(omega))
        return real_if_real(plan.download(0, SLOT_V), v0, f, A).reshape(n, 1)

    def _vcycle_general(self, v0, f, A, kind, omega, nu1, nu2, nu_coarse, shift, lowest_level):
        """vcycle (MGCMTSolver.py:281-329) for an operator ``recognise`` cannot map — any square scipy.sparse matrix, real
        or complex, cycled as ONE 1-D grid of its full length (ThesisProblem.py:101: a complex 4n x 4n block matrix,
        smoother=solver.gseidel, lowest_level=2**5).  Galerkin hierarchy, smoothing, transfers and the coarsest solve
        run on the GPU (csrc/csr.hip); the result is complex when an input is."""
        from .general import get_csr_plan, real_if_real
        if kind is None:
            raise NotImplementedError("general sparse operators take this class's wjacobi / gseidel / sor")
        if np.iscomplexobj(shift):
            raise NotImplementedError("complex shifts are not supported (the reference's are real guesses of eigenvalues)")
        n = len(v0)
        plan = get_csr_plan(A, int(lowest_level))
        plan.upload(0, SLOT_V, v0)
        plan.upload(0, SLOT_F, f)
        plan.vcycle(nu1, nu2, kind, omega=omega, nu_coarse=nu_coarse, shift=float(shift))
        v = real_if_real(plan.download(0, SLOT_V), v0, f, A)
        return v.reshape(n, 1) if n == lowest_level else v

    def wjacobi(self, v0, f, A, nu=4, omega=2. / 3.):
        """MGCMTSolver.py:182-208 — v <- (I - w D^-1 A) v + w D^-1 f, nu times; returns (n, 1)."""
        n = len(v0)
        _force_column(f, n)
        _force_column(v0, n)
        return self._smooth(v0, f, A, WJACOBI, nu, omega)

    def gseidel(self, v0, f, A, nu=4):
        """MGCMTSolver.py:210-227 — forward Gauss-Seidel in index order; returns (n, 1).  (The
        reference only works for (n, 1) inputs; 1-D inputs are treated as columns here.)"""
        return self._smooth(v0, f, A, GS_LEX, nu, 1.0)

    def sor(self, v0, f, A, nu=4, omega=1):
        """MGCMTSolver.py:229-246 — v <- (D-wL)^-1((1-w)D + wU) v + w (D-L)^-1 f, including the
        reference's (D-L)^-1 on the right-hand side (:241); returns (n, 1)."""
        return self._smooth(v0, f, A, SOR_LEX, nu, omega)

    def gseidel_rb(self, v0, f, A, nu=4, omega=1.0, dimension=None):
        """Addition — the red-black smoother the reference left as "TODO: FIX GSEIDELRB"
        (MGCMTSolver.py:248-279): odd indices first in 1-D; in 2-D the colours (i%2, j%2) in the order
        (0,1),(1,0),(0,0),(1,1), i.e. red-black on 5-point operators and a proper four-colour
        Gauss-Seidel on the 9-point Galerkin operators of the coarse levels."""
        return self._smooth(v0, f, A, GS_MC, nu, omega, dimension=dimension)

    def smooth(self, v0, f, A, nu=4, smoother=None, dimension=None):
        """Addition (named by the build's north star): dispatch on a smoother callable."""
        kind, omega = self._resolve_smoother(smoother)
        if kind is None:                       # a foreign callable: it is the smoother
            return omega(v0, f, A, nu=nu)
        return self._smooth(v0, f, A, kind, nu, omega, dimension=dimension)

    def _resolve_smoother(self, smoother):
        if smoother is None:
            return WJACOBI, 2. / 3.
        func, kw = smoother, {}
        if isinstance(smoother, functools.partial):
            func, kw = smoother.func, dict(smoother.keywords or {})
        owner, name = getattr(func, "__self__", None), getattr(func, "__name__", "")
        if isinstance(owner, MGCMTSolver):
            if name == "wjacobi":
                return WJACOBI, float(kw.get("omega", 2. / 3.))
            if name == "gseidel":
                return GS_LEX, 1.0
            if name == "sor":
                return SOR_LEX, float(kw.get("omega", 1))
            if name == "gseidel_rb":
                return GS_MC, float(kw.get("omega", 1.0))
        if callable(smoother):
            return None, smoother                  # seam 1 (MGCMTSolver.py:313,326): a foreign callable, run on host arrays
        raise TypeError("smoother= must be callable as smoother(v0, f, shifted_matrix, nu=...)")

    # -- the reference's two injection seams ---------------------------------------------------------
    _checked_stencil_makers = {}

    def _check_stencil_maker(self, stencil_maker, dimension):
        """Seam 2 (MGCMTSolver.py:310-311): the cycle's grid transfers are the kernels' built-in full weighting and
        (bi)linear interpolation — the matrices of MGCMTStencilMaker.py:27-78.  A caller's own stencil maker is accepted
        when its matrices ARE those (checked once per class on a small grid); anything else raises instead of being
        silently replaced."""
        if stencil_maker is None or isinstance(stencil_maker, MGCMTStencilMaker):
            return
        key = (type(stencil_maker), dimension)
        ok = self._checked_stencil_makers.get(key)
        if ok is None:
            own = self.stencil_maker
            try:
                ok = True
                for coarse, fine in ((4, 8), (8, 16)):
                    R, P = stencil_maker.restriction(fine, coarse, dimension=dimension), stencil_maker.interpolation(coarse, fine, dimension=dimension)
                    R0, P0 = own.restriction(fine, coarse, dimension=dimension), own.interpolation(coarse, fine, dimension=dimension)
                    for got, want in ((R, R0), (P, P0)):
                        got = got.toarray() if hasattr(got, "toarray") else np.asarray(got)
                        ok = ok and got.shape == want.shape and np.abs(got - want.toarray()).max() <= 1e-14
            except Exception:
                ok = False
            self._checked_stencil_makers[key] = ok
        if not ok:
            raise ValueError(
                "stencil_maker=%r builds restriction / interpolation matrices that differ from full weighting / (bi)linear "
                "interpolation (MGCMTStencilMaker.py:27-78); the HIP V-cycle has exactly those transfers built in and will "
                "not silently substitute them" % (stencil_maker,))

    def _level_matrix(self, plan, level, shift):
        """(A_level - shift I) as the scipy.sparse matrix the reference hands to a smoother (:287-288,313); A_level is
        the Galerkin operator R*A*P of that level (:318), rebuilt from the plan's Kronecker factors."""
        xf = plan.factors(level, 0) if plan.dim == 2 else None
        yf = plan.factors(level, 1)
        terms = [((xf[m].copy() if xf is not None else None), yf[m].copy()) for m in range(yf.shape[0])]
        op = StructuredOperator("2d" if plan.dim == 2 else "1d", plan.g >> level, terms)
        if shift:
            op = op.shifted(float(shift))
        return tag_structured(op.tocsr(), op)      # this class's own smoothers map it straight back (9-point levels too)

    def _cycle_with_host_smoother(self, plan, smoother, nu1, nu2, nu_coarse, k, shifts, gram_schmidt, positional_nu=False):
        """The cycle of vcycle / vcycle_matrix with a FOREIGN smoother callable: residual, restriction, coarse solve,
        interpolation + correction and Gram-Schmidt stay on the device; on every level the iterate and the right-hand
        side visit the host, where the caller's function is applied exactly as the reference applies it —
        ``smoother(v, f, shifted_matrix, nu=nu)`` with (n, 1) arrays and the level's sparse matrix (:313,326,416,432)."""
        last = plan.num_levels - 1
        mats = {}

        def smooth(level, nu):
            n = plan.size(level)
            for q in range(k):
                key = (level, float(shifts[q]))
                if key not in mats:
                    mats[key] = self._level_matrix(plan, level, shifts[q])
                v = plan.download(level, SLOT_V, q).reshape(n, 1)
                f = plan.download(level, SLOT_F, q).reshape(n, 1)
                out = np.asarray(smoother(v, f, mats[key], nu) if positional_nu else smoother(v, f, mats[key], nu=nu), dtype=np.float64)
                if out.size != n:
                    raise ValueError("smoother returned %r values for a level of %d" % (out.shape, n))
                plan.upload(level, SLOT_V, q, out.reshape(-1))

        for l in range(last):
            smooth(l, nu1 if l == 0 else nu_coarse)
            plan.residual_restrict(l, k=k)              # F[l+1] = R (F - (A - mu) V), V[l+1] = 0   (:315-316)
        plan.coarse_solve(last, k=k)
        for l in range(last - 1, -1, -1):
            plan.prolong_correct(l, k=k)                # V += P V[l+1]                              (:323-324)
            smooth(l, nu2 if l == 0 else nu_coarse)
            if gram_schmidt:
                plan.gramschmidt(l, SLOT_V, k, modified=1)

    # ------------------------------------------------------------------------------------------
    # cycles
    # ------------------------------------------------------------------------------------------
    def _grid(self, n, dimension):
        if dimension == "1d":
            return n
        if dimension == "2d":
            return np.sqrt(n)
        return 0

    def _check_grid(self, g, lowest_level):
        """True when the cycle can run; otherwise the reference's message is printed."""
        if g < 2:
            print("Length of start vector is not a power of 2")
            return False
        if int(g) != g or not _is_pow2(g):
            print("Old gridsize isn't a power of 2 !")     # what stencil_maker.restriction prints (:72)
            return False
        if lowest_level > g or not _is_pow2(lowest_level) or lowest_level < 2:
            raise ValueError("lowest_level=%r is never reached from a grid of %d" % (lowest_level, int(g)))
        return True

    def vcycle(self, v0, f, A, stencil_maker, nu1=4, nu2=4, smoother=None, shift=0, lowest_level=2, dimension="1d",
               *, nu_coarse=4):
        """MGCMTSolver.py:281-329.  One V-cycle for (A - shift I) v = f.

        Kept from the reference: the caller's v0/f get shape (n, 1) (:297-300); levels below the top
        run V(4,4) because nu1/nu2 are not forwarded (:320) — ``nu_coarse`` (addition) overrides the 4;
        the Galerkin operator is built from the unshifted A and the shift re-applied as -shift*I on
        every level (:287-288,318); the result is 1-D (:329) except when the start grid already is the
        lowest level, where it is (n, 1) (:305-308); bad sizes print and return None (:303-304).
        ``stencil_maker``: the transfer operators are the kernels' built-in full weighting / (bi)linear interpolation
        (MGCMTStencilMaker.py:27-78); an object whose matrices differ raises (``_check_stencil_maker``).
        ``smoother``: one of this class's methods runs inside the device cycle; any other callable is applied on host
        arrays level by level with the transfers on the device (``_cycle_with_host_smoother``).
        """
        kind, omega = self._resolve_smoother(smoother)
        self._check_stencil_maker(stencil_maker, dimension)
        n = len(v0)
        g = self._grid(n, dimension)
        _force_column(f, n)
        _force_column(v0, n)
        if not self._check_grid(g, lowest_level):
            return None
        g = int(g)
        try:
            op = recognise(A, dimension)
        except UnrecognisedOperator:
            if dimension != "1d":
                raise
            return self._vcycle_general(v0, f, A, kind, omega, int(nu1), int(nu2), int(nu_coarse), shift, int(lowest_level))
        plan = get_plan(op, int(lowest_level), nvec=1)
        plan.set_shifts([float(shift)])
        v0 = np.asarray(v0, dtype=np.float64).reshape(-1)
        zero_start = _all_zero(v0)                # a zero start vector (the reference's drivers) is a flag, not a transfer
        if not zero_start:
            plan.upload(0, SLOT_V, 0, v0)
        elif kind is None:
            plan.zero(0, SLOT_V, 0)
        plan.upload(0, SLOT_F, 0, np.asarray(f, dtype=np.float64).reshape(-1))
        if kind is None:
            self._cycle_with_host_smoother(plan, omega, int(nu1), int(nu2), int(nu_coarse), 1, [float(shift)], False)
        else:
            plan.vcycle(int(nu1), int(nu2), kind, omega=omega, k=1, nu_coarse=int(nu_coarse), zero_start=zero_start)
        v = plan.download(0, SLOT_V, 0)
        if g == lowest_level:
            return v.reshape(n, 1)
        return v

    def fmg(self, f, A, stencil_maker, nu1=4, nu2=4, smoother=None, shift=0, lowest_level=2, dimension="1d", *,
            nu_coarse=4, cycles_per_level=1):
        """Full multigrid / nested iteration (an ADDITION: the reference's report describes it as the next step — PDF
        p.17 "FMG", p.51 — but its code has no such function).  The right-hand side is restricted through all levels
        (full weighting, the R of MGCMTStencilMaker.py:57-78), the coarsest problem is solved directly, and on the way
        up every level takes the interpolated coarser solution as its start value and runs ``cycles_per_level``
        V-cycles (the ones of ``vcycle``: V(nu1,nu2) on their top level, V(nu_coarse,nu_coarse) below) on the
        Galerkin operator of that level.  Returns the fine-grid solution; about 4/3 (2-D) of the work of one V-cycle
        per cycle and level.  Built from the same C-ABI calls as ``vcycle``."""
        kind, omega = self._resolve_smoother(smoother)
        if kind is None:
            raise NotImplementedError("fmg (an addition, not a reference function) takes this class's smoothers only")
        self._check_stencil_maker(stencil_maker, dimension)
        n = len(f)
        g = self._grid(n, dimension)
        if not self._check_grid(g, lowest_level):
            return None
        op = recognise(A, dimension)
        plan = get_plan(op, int(lowest_level), nvec=1)
        plan.set_shifts([float(shift)])
        V, F = (SLOT_V, 0), (SLOT_F, 0)
        plan.upload(0, SLOT_F, 0, np.asarray(f, dtype=np.float64).reshape(-1))
        last = plan.num_levels - 1
        for l in range(last):
            plan.restrict(l, F, F)                                  # f_{l+1} = R f_l
        plan.coarse_solve(last)
        for l in range(last - 1, -1, -1):
            plan.prolong(l, V, V, accumulate=False)                 # start value: P v_{l+1}
            for _ in range(int(cycles_per_level)):
                plan.vcycle(int(nu1), int(nu2), kind, omega=omega, k=1, nu_coarse=int(nu_coarse), level=l)
        return plan.download(0, SLOT_V, 0)

    def twogrid(self, v0, f, A, stencil_maker, nu1=4, nu2=4, smoother=None, shift=0, dimension="1d"):
        """MGCMTSolver.py:331-371 — pre-smooth, exact solve of (R A P - shift I) on the next grid,
        post-smooth.  (The reference sizes the coarse shift as n/2 (:350) and therefore only runs in
        1-D; here 2-D works as well.)"""
        kind, omega = self._resolve_smoother(smoother)
        n = len(v0)
        g = self._grid(n, dimension)
        _force_column(f, n)
        _force_column(v0, n)
        if not self._check_grid(g, 2):
            return None
        self._check_stencil_maker(stencil_maker, dimension)
        g = int(g)
        if g < 4:
            raise ValueError("twogrid needs a fine grid of at least 4 points per direction")
        op = recognise(A, dimension)
        plan = get_plan(op, g // 2, nvec=1)
        plan.set_shifts([float(shift)])
        plan.upload(0, SLOT_V, 0, np.asarray(v0, dtype=np.float64).reshape(-1))
        plan.upload(0, SLOT_F, 0, np.asarray(f, dtype=np.float64).reshape(-1))
        if kind is None:                       # the reference passes nu positionally here (:358,369)
            self._cycle_with_host_smoother(plan, omega, int(nu1), int(nu2), 4, 1, [float(shift)], False, positional_nu=True)
        else:
            plan.twogrid(int(nu1), int(nu2), kind, omega=omega, k=1)
        return plan.download(0, SLOT_V, 0)

    def vcycle_matrix(self, v0_matrix, f_matrix, A, stencil_maker, nu1=4, nu2=4, smoother=None, shifts=None,
                      lowest_level=2, dimension="1d", *, nu_coarse=4):
        """MGCMTSolver.py:375-436 — the V-cycle on k columns at once, one shift per column, modified
        Gram-Schmidt of the columns on every non-coarsest level on the way up (:434).  All columns move
        through every kernel together (grid z = column).  ``shifts=None`` means zero shifts (the
        reference's default crashes)."""
        kind, omega = self._resolve_smoother(smoother)
        v0_matrix = np.asarray(v0_matrix, dtype=np.float64)
        f_matrix = np.asarray(f_matrix, dtype=np.float64)
        n = len(v0_matrix[:, 0])
        k = f_matrix.shape[1]
        shifts = np.zeros(k) if shifts is None else np.asarray(shifts, dtype=np.float64).reshape(-1)
        if len(shifts) != k:
            raise ValueError("dimension mismatch: %d shifts for %d columns" % (len(shifts), k))
        if k > _lib.MAX_VEC:
            raise ValueError("at most %d columns per call" % _lib.MAX_VEC)
        g = self._grid(n, dimension)
        if not self._check_grid(g, lowest_level):
            return None
        self._check_stencil_maker(stencil_maker, dimension)
        op = recognise(A, dimension)
        plan = get_plan(op, int(lowest_level), nvec=k)
        plan.set_shifts(shifts)
        zero_start = _all_zero(v0_matrix)         # the reference's callers pass zeros (1DPotMatrixVcycle.py:70): nothing to upload
        for i in range(k):
            if not zero_start:
                plan.upload(0, SLOT_V, i, v0_matrix[:, i])
            plan.upload(0, SLOT_F, i, f_matrix[:, i])
        if kind is None:
            if zero_start:
                for i in range(k):
                    plan.zero(0, SLOT_V, i)
            self._cycle_with_host_smoother(plan, omega, int(nu1), int(nu2), int(nu_coarse), k, shifts, True)
        else:
            plan.vcycle(int(nu1), int(nu2), kind, omega=omega, k=k, nu_coarse=int(nu_coarse), gram_schmidt=True, zero_start=zero_start)
        # columns are downloaded as contiguous rows of a (k, n) array; the (n, k) result is its transpose (a
        # column-major array: the same values and indexing as the reference's, without k strided scatters on the host)
        rows = np.empty((k, n))
        for i in range(k):
            plan.download_into(0, SLOT_V, i, rows[i])
        return rows.T

    # ------------------------------------------------------------------------------------------
    # grid transfers as stand-alone calls (addition; the stale drivers main.py:81,163 and
    # shiftMethod.py:77 call solver.interpolate(coarse_vec, stencil_maker, new_gridsize))
    # ------------------------------------------------------------------------------------------
    def interpolate(self, vec, stencil_maker, new_gridsize, dimension="1d"):
        """stencil_maker.interpolation(old, new) * vec, computed by chaining the two-level
        prolongation kernel (the multi-level matrix equals the product of two-level ones)."""
        vec = np.asarray(vec, dtype=np.float64).reshape(-1)
        old = len(vec) if dimension == "1d" else int(round(math.sqrt(len(vec))))
        new = int(new_gridsize)
        if not (_is_pow2(old) and _is_pow2(new) and new > old):
            print("New gridsize isn't bigger than old gridsize !" if new <= old else "Old gridsize isn't a power of 2 !")
            return None
        plan = get_plan(laplacian_operator(new, dimension), old, nvec=1)
        last = plan.num_levels - 1
        plan.upload(last, SLOT_V, 0, vec)
        for l in range(last - 1, -1, -1):
            plan.prolong(l, (SLOT_V, 0), (SLOT_V, 0), accumulate=False)
        return plan.download(0, SLOT_V, 0)

    def restrict(self, vec, stencil_maker, new_gridsize, dimension="1d"):
        """stencil_maker.restriction(old, new) * vec by chaining the full-weighting kernel; in 2-D
        the reference's fixed 1/4 prefactor for multi-level jumps (MGCMTStencilMaker.py:77) is kept."""
        vec = np.asarray(vec, dtype=np.float64).reshape(-1)
        old = len(vec) if dimension == "1d" else int(round(math.sqrt(len(vec))))
        new = int(new_gridsize)
        if not (_is_pow2(old) and _is_pow2(new) and new < old):
            print("New gridsize is bigger (more elements) than old gridsize !" if new >= old else "Old gridsize isn't a power of 2 !")
            return None
        plan = get_plan(laplacian_operator(old, dimension), new, nvec=1)
        plan.upload(0, SLOT_V, 0, vec)
        for l in range(plan.num_levels - 1):
            plan.restrict(l, (SLOT_V, 0), (SLOT_V, 0))
        out = plan.download(plan.num_levels - 1, SLOT_V, 0)
        if dimension == "2d" and plan.num_levels > 2:
            out *= 4.0 ** (plan.num_levels - 2)
        return out

    # ------------------------------------------------------------------------------------------
    # Rayleigh-quotient minimisation (MGCMTSolver.py:17-122)
    # ------------------------------------------------------------------------------------------
    _X, _P, _G, _GOLD, _AX, _AP, _MX, _MP, _TMP, _TMP2 = range(10)
    _RQ_REGS = 10

    def _rqmin_device(self, plan, level, nu, robust=False, want_rho=True):
        """rqmin (MGCMTSolver.py:17-57) on the registers of `level`; x is register _X.  Resident on the device
        (mgcmt_rqmin, csrc/kernels_rq.hip): per step two passes over the data — p = -g + beta p_old formed on the fly, A and M
        applied to x and p in registers with the eight inner products of :33-46; then x + delta p with its gradient and the
        next step's inner products — the 2 x 2 generalised eigenproblem of :48-50 solved in closed form by one workgroup,
        and no host round trip (the round-2 form took six operator applications, three Gram passes, four vector updates
        and three round trips per step: ~300 B per point against 64).  ``robust`` (the repaired variants, (f)4): a
        degenerate pencil — the search direction vanished or is parallel to x, as happens with two orthogonal columns on a
        2-point grid, where the reference dies inside eig with "array must not contain infs or NaNs"  — ends
        the minimisation on this level.  want_rho=False: nothing synchronises."""
        return plan.rqmin(level, SLOT_V, self._rq_vecs(), int(nu), robust=robust, want_rho=want_rho)

    def _rq_vecs(self):
        """registers of mgcmt_rqmin: the iterate, its ping-pong partner, p and its partner, g, one for M g"""
        return [self._X, self._TMP, self._P, self._TMP2, self._G, self._GOLD]

    def _rq_plan(self, A, M, nmin, dimension=None):
        if M is None:
            raise AttributeError("'NoneType' object has no attribute 'dot'")      # what the reference raises (:19)
        opA = recognise(A, dimension)
        opM = recognise(M, opA.dimension)
        return get_plan(opA, int(nmin), nvec=self._RQ_REGS, mass=opM)

    def rqmin(self, A, v0, M=None, nu=4):
        """MGCMTSolver.py:17-57 — returns (x, rho).  (The reference's x / rho are complex-typed with
        zero imaginary part because scipy's eig returns complex; real values are returned here.)"""
        if M is None:
            raise AttributeError("'NoneType' object has no attribute 'dot'")      # what the reference raises (:19)
        x0 = np.asarray(v0, dtype=np.float64).reshape(-1)
        opA = recognise(A)
        plan = get_plan(opA, opA.g, nvec=self._RQ_REGS, mass=recognise(M, opA.dimension))
        plan.set_shifts(np.zeros(self._RQ_REGS))
        plan.upload(0, SLOT_V, self._X, x0)
        rho = self._rqmin_device(plan, 0, int(nu))
        return plan.download(0, SLOT_V, self._X), rho

    def _rqmg_levels(self, plan, level, nu1, nu2, robust=False):
        """vcycle_rqmg's recursion (MGCMTSolver.py:99-122) from `level` down.  From the finest level the whole cycle is one
        call into the library (mgcmt_vcycle_rqmg: no host round trip, replayed as a HIP graph); the host sees one number
        per cycle, the Rayleigh quotient of the finest level's last minimisation."""
        if level == 0:
            return None, plan.vcycle_rqmg(SLOT_V, self._rq_vecs(), nu1, nu2, robust=robust)
        last = level + 1 >= plan.num_levels
        rho = self._rqmin_device(plan, level, nu1, robust, want_rho=False)
        if not last:
            plan.restrict(level, (SLOT_V, self._X), (SLOT_V, self._X))             # k_coarse = R k   (:113)
            self._rqmg_levels(plan, level + 1, nu1, nu2, robust)
            plan.prolong(level, (SLOT_V, self._X), (SLOT_V, self._X), accumulate=True)  # k += P c   (:116-118)
            rho = self._rqmin_device(plan, level, nu2, robust, want_rho=False)
        return None, rho

    def vcycle_rqmg(self, x, A, M, nu1=4, nu2=4, nmin=2):
        """MGCMTSolver.py:99-122 — Rayleigh-quotient multigrid: rqmin, restrict the ITERATE, recurse
        on the Galerkin pair (R A P, R M P), add the interpolated coarse iterate, rqmin; returns (k, rho).
        The reference only has the 1-D transfer operators here (:107-108); for a 2-D operator (a
        StructuredOperator) the same algorithm runs with the 2-D ones and nmin counts points per direction."""
        x0 = np.asarray(x, dtype=np.float64).reshape(-1)
        g = recognise(A).g                     # points per direction (the vector length in 1-D, as in the reference)
        plan = self._rq_plan(A, M, max(int(nmin), 2) if g > nmin else g)
        plan.set_shifts(np.zeros(self._RQ_REGS))
        plan.upload(0, SLOT_V, self._X, x0)
        _, rho = self._rqmg_levels(plan, 0, int(nu1), int(nu2))
        return plan.download(0, SLOT_V, self._X), rho

    def vcycle_rqmg2(self, x_matrix, A, M, nu1=4, nu2=4, nmin=2, level=0, *, repaired=False):
        """MGCMTSolver.py:59-94 — multi-vector variant: rqmin per column, four Gram-Schmidt passes on
        the finest level (:69-71), recursion on the restricted columns, correction and rqmin per column.
        ``repaired`` (addition): with the default nmin=2 the reference dies inside ``eig`` (two
        orthogonal columns degenerate on a 2-point grid); repaired=True ends a level's minimisation when its 2x2
        pencil degenerates and solves it with ``eigh`` on the symmetrised matrices, so the default arguments run."""
        k0 = np.array(x_matrix, dtype=np.float64)
        n, nv = k0.shape
        if nv > _lib.MAX_VEC:
            raise ValueError("at most %d columns per call" % _lib.MAX_VEC)
        opA = recognise(A)
        plan = get_plan(opA, max(int(nmin), 2) if opA.g > nmin else opA.g, nvec=max(self._RQ_REGS, nv), mass=recognise(M, opA.dimension))
        plan.set_shifts(np.zeros(plan.nvec))
        for i in range(nv):
            plan.upload(0, SLOT_W, i, k0[:, i])
        self._rqmg2_levels(plan, 0, nv, int(nu1), int(nu2), bool(repaired))
        out = np.zeros((n, nv))
        for i in range(nv):
            out[:, i] = plan.download(0, SLOT_W, i)
        return out

    def _rqmg2_levels(self, plan, level, nv, nu1, nu2, robust=False):
        for i in range(nv):
            plan.copy(level, SLOT_W, i, SLOT_V, self._X)
            self._rqmin_device(plan, level, nu1, robust, want_rho=False)
            plan.copy(level, SLOT_V, self._X, SLOT_W, i)
        if level == 0:
            for _ in range(4):
                plan.gramschmidt(level, SLOT_W, nv, modified=1)
        if level + 1 < plan.num_levels:
            for i in range(nv):
                plan.restrict(level, (SLOT_W, i), (SLOT_W, i))
            self._rqmg2_levels(plan, level + 1, nv, nu1, nu2, robust)
            for i in range(nv):
                plan.prolong(level, (SLOT_W, i), (SLOT_W, i), accumulate=True)
                plan.copy(level, SLOT_W, i, SLOT_V, self._X)
                self._rqmin_device(plan, level, nu2, robust, want_rho=False)
                plan.copy(level, SLOT_V, self._X, SLOT_W, i)

    def twogridrqmin(self, A, v0, M, nu1=4, nu2=4, *, repaired=False):
        """MGCMTSolver.py:127-178 is dead code in the reference: it calls ``eigh`` which is never imported (:167 vs :4)
        and raises NameError on first use — reproduced by default.

        ``repaired=True`` (addition) runs what the function is the two-grid form of: the reference's own
        Rayleigh-quotient multigrid (vcycle_rqmg, :99-122) cut off after one coarsening — rqmin on the fine grid (:133),
        the iterate restricted (:141), the coarse pair (R A P, R M P) minimised with nu1 rqmin steps (the loop of
        :151-174, whose x_coarse the dead code never updates), the interpolated coarse iterate added (:170) and nu2
        fine-grid rqmin steps (:176).  Equal to ``vcycle_rqmg(v0, A, M, nu1, nu2, nmin=n/2)`` of the reference
        (fixture tests/golden/rqmin.npz: twogrid_*)."""
        if not repaired:
            raise NameError("name 'eigh' is not defined")
        x0 = np.asarray(v0, dtype=np.float64).reshape(-1)
        g = recognise(A).g
        if g < 4:
            raise ValueError("twogridrqmin needs at least 4 points per direction")
        plan = self._rq_plan(A, M, g // 2)
        plan.set_shifts(np.zeros(self._RQ_REGS))
        plan.upload(0, SLOT_V, self._X, x0)
        _, rho = self._rqmg_levels(plan, 0, int(nu1), int(nu2), robust=True)
        return plan.download(0, SLOT_V, self._X), rho
