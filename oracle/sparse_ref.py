"""Generic-sparse CPU restatement of the reference's multigrid path (NumPy/SciPy, Python 3).

TEST INFRASTRUCTURE — NOT PRODUCT CODE (see oracle/__init__.py).

Every function cites the reference lines it restates (paths relative to /root/reference).  The
semantics (including the reference's quirks) are kept; the cost is not: the reference
materialises each smoother's iteration matrix with N sparse solves (MGCMTSolver.py:195,215,234),
here a sweep is one sparse mat-vec / one triangular solve.

Parity: pinned by tests/test_oracle_golden.py against the reference's own known-answer values and
against tests/golden/*.npz written by oracle/gen_golden.py from the reference itself.
"""
import math

import numpy as np
import scipy.linalg
import scipy.sparse as sp
from scipy.sparse.linalg import spsolve, spsolve_triangular


# ----------------------------------------------------------------------------------------------
# operators  (MGCMTStencilMaker.py)
# ----------------------------------------------------------------------------------------------

def _log2_int(x):
    p = math.log(x) / math.log(2)
    return p, float(p).is_integer()


class RefStencilMaker:
    def laplacian(self, n, dimension="1d"):
        """MGCMTStencilMaker.py:15-25 — 1-D tridiag(1,-2,1)/h^2 with h = 1/n; 2-D kronsum."""
        n = int(n)
        h = 1.0 / n
        if dimension == "1d":
            lap = sp.diags([1.0, -2.0, 1.0], [-1, 0, 1], shape=(n, n), format="csc")
            return lap * (1 / h ** 2)
        if dimension == "2d":
            one_d = self.laplacian(n, "1d")
            return sp.kronsum(one_d, one_d).tocsr()
        return None

    def interpolation(self, old_gridsize, new_gridsize, dimension="1d"):
        """MGCMTStencilMaker.py:27-54 — hat functions of half-width m = new/old centred on the fine
        index (J+1)*m - 1, scaled by 1/m; 2-D is kron(S, S).  Bad sizes print and return None."""
        new_gridsize = int(new_gridsize)
        if dimension == "2d":
            s = self.interpolation(old_gridsize, new_gridsize, "1d")
            return None if s is None else sp.kron(s, s, format="csc")
        p_old, old_ok = _log2_int(old_gridsize)
        p_new, new_ok = _log2_int(new_gridsize)
        if not p_new > p_old:
            print("New gridsize isn't bigger than old gridsize !")
            return None
        if not old_ok:
            print("Old gridsize isn't a power of 2 !")
            return None
        if not new_ok:
            print("New gridsize isn't a power of 2 !")
            return None
        old = int(old_gridsize)
        m = new_gridsize // old
        rows, cols, vals = [], [], []
        for j in range(old):
            centre = (j + 1) * m - 1
            for d in range(-(m - 1), m):
                r = centre + d
                if 0 <= r < new_gridsize:
                    rows.append(r)
                    cols.append(j)
                    vals.append((m - abs(d)) / float(m))
        return sp.csc_matrix((vals, (rows, cols)), shape=(new_gridsize, old))

    def restriction(self, old_gridsize, new_gridsize, dimension="1d"):
        """MGCMTStencilMaker.py:57-78 — 1-D (1/2)^p P^T; 2-D 1/4 kron(S,S)^T (the 1/4 is fixed even
        for multi-level jumps, MGCMTStencilMaker.py:77)."""
        if dimension == "2d":
            p = self.interpolation(new_gridsize, old_gridsize, "2d")
            return None if p is None else (0.25 * p.T).tocsr()
        p_old, old_ok = _log2_int(old_gridsize)
        p_new, new_ok = _log2_int(new_gridsize)
        if not p_new < p_old:
            print("New gridsize is bigger (more elements) than old gridsize !")
            return None
        if not old_ok:
            print("Old gridsize isn't a power of 2 !")
            return None
        if not new_ok:
            print("New gridsize isn't a power of 2 !")
            return None
        p = self.interpolation(new_gridsize, old_gridsize, "1d")
        return sp.csc_matrix((0.5 ** (p_old - p_new)) * p.T)


# ----------------------------------------------------------------------------------------------
# vectors  (MGCMTProcessor.py)
# ----------------------------------------------------------------------------------------------

class RefProcessor:
    def projection(self, v, u):
        """MGCMTProcessor.py:10-20 — (<v,u>/<u,u>) u."""
        return (float(np.inner(v, u)) / float(np.inner(u, u))) * u

    def gramschmidt(self, vectors, modified=1):
        """MGCMTProcessor.py:22-50 — classical (modified=0) or modified Gram-Schmidt on columns."""
        a = np.array(vectors, dtype=float)
        n, k = a.shape
        if not modified:
            q = np.zeros((n, k))
            for j in range(k):
                col = a[:, j].copy()
                acc = col.copy()
                for i in range(j):
                    acc = acc - self.projection(col, q[:, i])
                q[:, j] = acc
            return self.normalize(q)
        out = np.zeros((n, k))
        for i in range(k):
            out[:, i] = a[:, i] / np.linalg.norm(a[:, i])
            for j in range(i + 1, k):
                a[:, j] = a[:, j] - self.projection(a[:, j], out[:, i])
        return out

    def normalize(self, vectors):
        """MGCMTProcessor.py:52-63."""
        v = np.asarray(vectors, dtype=float)
        out = np.zeros(v.shape)
        for j in range(v.shape[1]):
            out[:, j] = v[:, j] / np.linalg.norm(v[:, j])
        return out

    def orthogonality_check(self, vectors):
        """MGCMTProcessor.py:65-73 — Gram matrix."""
        v = np.asarray(vectors)
        k = v.shape[1]
        g = np.zeros((k, k))
        for i in range(k):
            for j in range(k):
                g[i, j] = np.inner(v[:, i], v[:, j])
        return g


# ----------------------------------------------------------------------------------------------
# solver  (MGCMTSolver.py)
# ----------------------------------------------------------------------------------------------

def _col(x):
    a = np.asarray(x)
    return a.astype(complex if np.iscomplexobj(a) else float).reshape(-1).copy()


def colour_classes(n, dimension):
    """Index sets of the multicolour Gauss-Seidel ordering used by the performance mode.

    The reference has no working red-black smoother (``gseidelrb`` is commented out under
    "TODO: FIX GSEIDELRB", MGCMTSolver.py:248-279); its intent — odd 0-based indices first
    (:259-263,274) — fixes the convention: 1-D odd then even; 2-D four colours (i%2, j%2) in the
    order (0,1),(1,0),(0,0),(1,1), which for a 5-point operator is exactly red ((i+j) odd) then black.
    """
    idx = np.arange(n)
    if dimension == "1d":
        return [idx[idx % 2 == 1], idx[idx % 2 == 0]]
    g = int(round(math.sqrt(n)))
    i, j = idx // g, idx % g
    return [idx[(i % 2 == a) & (j % 2 == b)] for a, b in ((0, 1), (1, 0), (0, 0), (1, 1))]


class RefSolver:
    def __init__(self):
        self.stencil_maker = RefStencilMaker()
        self.processor = RefProcessor()

    # -- smoothers -----------------------------------------------------------------------------
    def wjacobi(self, v0, f, A, nu=4, omega=2. / 3.):
        """MGCMTSolver.py:182-208 — v <- (I - w D^-1 A) v + w D^-1 f, nu times."""
        v, f = _col(v0), _col(f)
        A = sp.csr_matrix(A)
        d = A.diagonal()
        for _ in range(nu):
            v = (v - omega * (A @ v) / d) + omega * (f / d)
        return v

    def wjacobi_iteration_matrix(self, v0, f, A, nu=4, omega=2. / 3.):
        """MGCMTSolver.py:193-206 with the reference's COST PROFILE as well as its result: the iteration matrix
        I - w D^-1 A is materialised by a sparse solve with a sparse right-hand side (N column solves, :195) and
        applied nu times; the right-hand-side term is solved again in every sweep (:202).  Same values as
        ``wjacobi`` (tests/test_oracle_golden.py); only bench.py's ``cpu_reference_equivalent`` uses it, to time
        what the reference's formulation costs on the box's host cores."""
        import warnings
        v, f = _col(v0), _col(f)
        n = v.shape[0]
        A = sp.csc_matrix(A)
        D = sp.diags(A.diagonal(), 0, shape=(n, n), format="csc")
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            R = sp.eye(n, format="csc") - omega * spsolve(D, A)
            for _ in range(nu):
                v = R @ v + omega * np.asarray(spsolve(D, f)).reshape(-1)
        return v

    def gseidel(self, v0, f, A, nu=4):
        """MGCMTSolver.py:210-227 — v <- (D-L)^-1 U v + (D-L)^-1 f: forward lexicographic GS."""
        return self.sor(v0, f, A, nu=nu, omega=1)

    def sor(self, v0, f, A, nu=4, omega=1):
        """MGCMTSolver.py:229-246 — v <- (D-wL)^-1((1-w)D + wU) v + w (D-L)^-1 f.  The right-hand
        side term is solved with (D-L), not (D-wL) (:241): reproduced."""
        v, f = _col(v0), _col(f)
        A = sp.csr_matrix(A)
        d = A.diagonal()
        strict_lower = sp.tril(A, -1, format="csr")
        minus_u = sp.triu(A, 1, format="csr")
        dl = (sp.diags(d) + strict_lower).tocsr()                 # D - L
        dwl = (sp.diags(d) + omega * strict_lower).tocsr()        # D - wL
        g = omega * spsolve_triangular(dl, f, lower=True)
        for _ in range(nu):
            rhs = (1 - omega) * d * v - omega * (minus_u @ v)
            v = spsolve_triangular(dwl, rhs, lower=True) + g
        return v

    def gseidel_mc(self, v0, f, A, nu=4, omega=1.0, dimension="1d"):
        """Multicolour (red-black / four-colour) Gauss-Seidel/SOR — the performance-mode smoother
        injected through the reference's ``smoother=`` seam (MGCMTSolver.py:281,313,326)."""
        v, f = _col(v0), _col(f)
        A = sp.csr_matrix(A)
        d = A.diagonal()
        for _ in range(nu):
            for c in colour_classes(len(v), dimension):
                r = f[c] - A[c, :] @ v
                v[c] = v[c] + omega * r / d[c]
        return v

    # -- cycles --------------------------------------------------------------------------------
    def _grid(self, n, dimension):
        return n if dimension == "1d" else np.sqrt(n)

    def vcycle(self, v0, f, A, stencil_maker, nu1=4, nu2=4, smoother=None, shift=0, lowest_level=2,
               dimension="1d"):
        """MGCMTSolver.py:281-329.  nu1/nu2 are NOT forwarded to the coarser levels (:320), which
        therefore run V(4,4); the Galerkin operator uses the unshifted A (:318) and the shift is
        re-applied as -shift*I on every level (:287-288)."""
        if smoother is None:
            smoother = self.wjacobi
        v0, f = _col(v0), _col(f)
        n = len(v0)
        shifted = sp.csr_matrix(A) - shift * sp.eye(n, format="csr")
        g = self._grid(n, dimension)
        if g < 2:
            print("Length of start vector is not a power of 2")
            return None
        if g == lowest_level:
            return np.asarray(spsolve(sp.csc_matrix(shifted), f)).reshape(-1)
        R = stencil_maker.restriction(g, g / 2, dimension=dimension)
        P = stencil_maker.interpolation(g / 2, g, dimension=dimension)
        v = _col(smoother(v0, f, shifted, nu=nu1))
        r = R @ (f - shifted @ v)
        coarse = R @ sp.csr_matrix(A) @ P
        e = self.vcycle(np.zeros(len(r)), r, coarse, stencil_maker, shift=shift, smoother=smoother,
                        lowest_level=lowest_level, dimension=dimension)
        v = v + P @ _col(e)
        return _col(smoother(v, f, shifted, nu=nu2))

    def twogrid(self, v0, f, A, stencil_maker, nu1=4, nu2=4, smoother=None, shift=0, dimension="1d"):
        """MGCMTSolver.py:331-371 — exact coarse solve with RAP - shift*I_{n/2} (1-D only, :350)."""
        if smoother is None:
            smoother = self.wjacobi
        v0, f = _col(v0), _col(f)
        n = len(v0)
        g = self._grid(n, dimension)
        shifted = sp.csr_matrix(A) - shift * sp.eye(n, format="csr")
        R = stencil_maker.restriction(g, g / 2, dimension=dimension)
        P = stencil_maker.interpolation(g / 2, g, dimension=dimension)
        coarse = R @ sp.csr_matrix(A) @ P - shift * sp.eye(n // 2)
        v = _col(smoother(v0, f, shifted, nu1))
        r = R @ (f - shifted @ v)
        e = np.asarray(spsolve(sp.csc_matrix(coarse), r)).reshape(-1)
        v = v + P @ e
        return _col(smoother(v, f, shifted, nu2))

    def vcycle_matrix(self, v0_matrix, f_matrix, A, stencil_maker, nu1=4, nu2=4, smoother=None,
                      shifts=None, lowest_level=2, dimension="1d"):
        """MGCMTSolver.py:375-436 — k columns, one shift per column, modified Gram-Schmidt on every
        non-coarsest level on the way up (:434); nu1/nu2 not forwarded (:426)."""
        if smoother is None:
            smoother = self.wjacobi
        v0_matrix = np.asarray(v0_matrix, dtype=float)
        f_matrix = np.asarray(f_matrix, dtype=float)
        n, k = v0_matrix.shape[0], f_matrix.shape[1]
        shifts = np.zeros(k) if shifts is None else np.asarray(shifts, dtype=float).reshape(-1)
        A = sp.csr_matrix(A)
        shifted = [A - s * sp.eye(n, format="csr") for s in shifts]
        g = self._grid(n, dimension)
        if g < 2:
            print("Length of start vector is not a power of 2")
            return None
        v = np.zeros((n, k))
        if g == lowest_level:
            for i in range(k):
                v[:, i] = np.asarray(spsolve(sp.csc_matrix(shifted[i]), f_matrix[:, i])).reshape(-1)
            return v
        R = stencil_maker.restriction(g, g / 2, dimension=dimension)
        P = stencil_maker.interpolation(g / 2, g, dimension=dimension)
        r = np.zeros((R.shape[0], k))
        for i in range(k):
            v[:, i] = _col(smoother(v0_matrix[:, i], f_matrix[:, i], shifted[i], nu=nu1))
            r[:, i] = R @ (f_matrix[:, i] - shifted[i] @ v[:, i])
        coarse = R @ A @ P
        e = self.vcycle_matrix(np.zeros(r.shape), r, coarse, stencil_maker, shifts=shifts,
                               smoother=smoother, lowest_level=lowest_level, dimension=dimension)
        for i in range(k):
            v[:, i] = v[:, i] + P @ e[:, i]
            v[:, i] = _col(smoother(v[:, i], f_matrix[:, i], shifted[i], nu=nu2))
        return self.processor.gramschmidt(v)

    # -- Rayleigh-quotient minimisation ----------------------------------------------------------
    # The 2x2 generalised eigenproblem of a step, solved as the reference solves it (scipy.linalg.eig = LAPACK's QZ, :48-50).
    # exact_pencil (NOT the reference; a test instrument): the same pencil solved in closed form in extended precision.
    # QZ is backward stable with respect to the NORM of the pencil, and p is of any size against x (a random start vector
    # on a fine grid: |p| ~ 1e7 |x|), so its delta = y1 / y0 carries a forward error of up to ~1e-9 relative that the
    # later steps inherit (measured: n = 8192, M = tridiag(1,4,1)/6).  The device code solves the pencil in closed form
    # (csrc/kernels_rq.hip) and agrees with the closed form to rounding; tests that go beyond the sizes of the reference's
    # own fixtures compare tightly with exact_pencil=True and loosely with the faithful form.
    exact_pencil = False

    @staticmethod
    def _pencil_delta_exact(Rm, RM):
        r00, r01, r10, r11 = (np.longdouble(v) for v in np.real(Rm).ravel())
        m00, m01, m10, m11 = (np.longdouble(v) for v in np.real(RM).ravel())
        a = m00 * m11 - m01 * m10
        b = -(r00 * m11 + m00 * r11) + (r01 * m10 + m01 * r10)
        c = r00 * r11 - r01 * r10
        disc = max(b * b - 4 * a * c, np.longdouble(0))
        q = -(b + (np.sqrt(disc) if b >= 0 else -np.sqrt(disc))) / 2
        roots = [q / a] + ([c / q] if q != 0 else [])
        lam = min(roots)
        return float(-(r10 - lam * m10) / (r11 - lam * m11))

    def rqmin(self, A, v0, M=None, nu=4):
        """MGCMTSolver.py:17-57 — CG-like Rayleigh-quotient minimisation; each step solves the 2x2
        generalised eigenproblem on span{x, p} (:38-51)."""
        x = np.array(v0, dtype=float).reshape(-1)
        rho = (x @ (A @ x)) / (x @ (M @ x))
        g = 2 * (A @ x - rho * (M @ x))
        g_old = x.copy()
        p = x.copy()
        for it in range(nu):
            if it == 0:
                p = -g
            else:
                p = -g + ((g @ (M @ g)) / (g_old @ (M @ g_old))) * p
            Ax, Ap, Mx, Mp = A @ x, A @ p, M @ x, M @ p
            Rm = np.array([[x @ Ax, x @ Ap], [p @ Ax, p @ Ap]])
            RM = np.array([[x @ Mx, x @ Mp], [p @ Mx, p @ Mp]])
            if self.exact_pencil:
                delta = self._pencil_delta_exact(Rm, RM)
            else:
                w, vecs = scipy.linalg.eig(Rm, b=RM)
                y = vecs[:, np.argmin(w)]
                delta = y[1] / y[0]
            x = x + delta * p
            rho = (x.conj() @ (A @ x)) / (x.conj() @ (M @ x))
            g_old = g
            g = 2 * (A @ x - rho * (M @ x))
        return x, rho

    def fmg(self, f, A, stencil_maker, nu1=4, nu2=4, smoother=None, shift=0, lowest_level=2, dimension="1d",
            cycles_per_level=1):
        """Full multigrid built from the reference's own pieces (NOT a reference function — PARITY UNPINNED; it is the
        oracle of MGCMTSolver.fmg, an addition): restrict f through the levels with `restriction`, solve on
        `lowest_level` as vcycle does (MGCMTSolver.py:305-308), then per level interpolate and run `vcycle` on the
        Galerkin operator of that level (:318)."""
        f = _col(f)
        g = len(f) if dimension == "1d" else int(round(math.sqrt(len(f))))
        ops, rhs, grids = [sp.csr_matrix(A)], [f], [g]
        while grids[-1] > lowest_level:
            gl = grids[-1]
            R = stencil_maker.restriction(gl, gl // 2, dimension=dimension)
            P = stencil_maker.interpolation(gl // 2, gl, dimension=dimension)
            ops.append(sp.csr_matrix(R @ ops[-1] @ P))
            rhs.append(R @ rhs[-1])
            grids.append(gl // 2)
        nl = ops[-1].shape[0]
        v = spsolve((ops[-1] - shift * sp.eye(nl)).tocsc(), rhs[-1]).reshape(-1)
        for l in range(len(ops) - 2, -1, -1):
            P = stencil_maker.interpolation(grids[l + 1], grids[l], dimension=dimension)
            v = P @ v
            for _ in range(cycles_per_level):
                v = self.vcycle(v, rhs[l], ops[l], stencil_maker, nu1=nu1, nu2=nu2, smoother=smoother, shift=shift,
                                lowest_level=lowest_level, dimension=dimension).reshape(-1)
        return v

    def vcycle_rqmg(self, x, A, M, nu1=4, nu2=4, nmin=2, dimension="1d"):
        """MGCMTSolver.py:99-122 — restricts the iterate itself, Galerkin A_c and M_c (:110-111).
        dimension="2d" (not in the reference, whose transfers here are 1-D only, :107-108) runs the same algorithm
        with the 2-D transfer operators; n then counts points per direction.  PARITY UNPINNED for "2d"."""
        k = np.array(x).reshape(-1)
        n = len(k) if dimension == "1d" else int(round(math.sqrt(len(k))))
        k, rho = self.rqmin(A, k, M, nu=nu1)
        if n > nmin:
            P = self.stencil_maker.interpolation(n // 2, n, dimension=dimension)
            R = self.stencil_maker.restriction(n, n // 2, dimension=dimension)
            c, rho = self.vcycle_rqmg(R @ k, R @ A @ P, R @ M @ P, nu1=nu1, nu2=nu2, nmin=nmin, dimension=dimension)
            k = k + P @ c
            k, rho = self.rqmin(A, k, M, nu=nu2)
        return k, rho

    def vcycle_rqmg2(self, x_matrix, A, M, nu1=4, nu2=4, nmin=2, level=0):
        """MGCMTSolver.py:59-94 — multi-vector variant, 4x Gram-Schmidt at level 0 (:69-71)."""
        k = np.array(x_matrix, dtype=float)
        n, nv = k.shape
        for i in range(nv):
            k[:, i], _ = self.rqmin(A, k[:, i], M, nu=nu1)
        if level == 0:
            for _ in range(4):
                k = self.processor.gramschmidt(k)
        if n > nmin:
            P = self.stencil_maker.interpolation(n // 2, n)
            R = self.stencil_maker.restriction(n, n // 2)
            kc = np.zeros((n // 2, nv))
            for i in range(nv):
                kc[:, i] = R @ k[:, i]
            c = self.vcycle_rqmg2(kc, R @ A @ P, R @ M @ P, nu1=nu1, nu2=nu2, nmin=nmin, level=level + 1)
            for i in range(nv):
                k[:, i] = k[:, i] + P @ c[:, i]
                k[:, i], _ = self.rqmin(A, k[:, i], M, nu=nu2)
        return k
