"""Structured description of the operators the reference hands to its solver.

The reference passes ``scipy.sparse`` matrices built by ``MGCMTStencilMaker.laplacian``
(MGCMTStencilMaker.py:15-25), usually pre-scaled — ``(-1 / np.pi ** 2) * laplacian``
(1DPotMatrixVcycle.py:16, 2DPotMatrixVcycle.py:27) — and sometimes pre-shifted (``A - mu*I`` handed to
a smoother, MGCMTSolver.py:313).  The kernels are matrix-free: they need the operator as a sum of
Kronecker products of tridiagonal factors,  A = sum_m X_m (x) Y_m  (rows (x) columns).  This module
recovers that form from a sparse matrix by an O(nnz) structural check (``recognise``) and offers a
matrix-free operator object (``StructuredOperator``) for grids too large to assemble.
"""
import hashlib
import math

import numpy as np
import scipy.sparse as sp


def tri_identity(n):
    t = np.zeros((3, n))
    t[1] = 1.0
    return t


def tri_laplacian(n):
    """MGCMTStencilMaker.py:17-21 — tridiag(1,-2,1) * (1/h**2), h = 1./n (same float expression)."""
    h = 1. / n
    s = 1 / h ** 2
    t = np.zeros((3, n))
    t[0, 1:] = 1.0 * s
    t[1] = -2.0 * s
    t[2, :-1] = 1.0 * s
    return t


def tri_to_sparse(t):
    n = t.shape[1]
    return sp.diags([t[0, 1:], t[1], t[2, :-1]], [-1, 0, 1], shape=(n, n), format="csr")


class StructuredOperator:
    """A = sum_m X_m (x) Y_m on a g x g grid (dimension "2d") or a single tridiagonal (``"1d"``).

    Behaves like the sparse matrix it stands for where the reference's callers need it: scalar
    ``*`` and ``/``, unary minus, ``.shape``, ``.diagonal()``, ``.tocsr()``/``.toarray()`` (small
    sizes), and ``A.dot(x)`` / ``A * x`` / ``A @ x`` which run the HIP apply kernel.
    """

    def __init__(self, dimension, g, terms):
        self.dimension = dimension
        self.g = int(g)
        self.terms = [(None if x is None else np.ascontiguousarray(x, dtype=np.float64),
                       np.ascontiguousarray(y, dtype=np.float64)) for x, y in terms]
        n = self.g if dimension == "1d" else self.g * self.g
        self.shape = (n, n)
        self._fingerprint = None

    # -- algebra with scalars ---------------------------------------------------------------------
    def _scaled(self, c):
        c = float(c)
        if self.dimension == "1d":
            return StructuredOperator("1d", self.g, [(None, y * c) for _, y in self.terms])
        return StructuredOperator("2d", self.g, [(x, y * c) for x, y in self.terms])

    def __mul__(self, other):
        if np.isscalar(other):
            return self._scaled(other)
        return self.dot(other)

    def __rmul__(self, other):
        if np.isscalar(other):
            return self._scaled(other)
        return NotImplemented

    def __truediv__(self, other):
        return self._scaled(1.0 / other)

    def __neg__(self):
        return self._scaled(-1.0)

    def __matmul__(self, other):
        return self.dot(other)

    def shifted(self, mu):
        """A - mu*I as a structured operator (the shift folded into the first term's diagonal)."""
        terms = [(None if x is None else x.copy(), y.copy()) for x, y in self.terms]
        if self.dimension == "1d":
            terms[0][1][1] -= mu
        else:
            terms.append((tri_identity(self.g), tri_identity(self.g) * (-float(mu))))
        return StructuredOperator(self.dimension, self.g, terms)

    # -- views ----------------------------------------------------------------------------------
    def diagonal(self):
        if self.dimension == "1d":
            return sum(y[1] for _, y in self.terms)
        return sum(np.outer(x[1], y[1]) for x, y in self.terms).reshape(-1)

    def tocsr(self):
        if self.shape[0] > (1 << 22):
            raise MemoryError("refusing to assemble a %d x %d sparse matrix" % self.shape)
        if self.dimension == "1d":
            return sum(tri_to_sparse(y) for _, y in self.terms).tocsr()
        return sum(sp.kron(tri_to_sparse(x), tri_to_sparse(y), format="csr") for x, y in self.terms).tocsr()

    def tocsc(self):
        return self.tocsr().tocsc()

    def toarray(self):
        return self.tocsr().toarray()

    def dot(self, x):
        from .plan import apply_operator
        return apply_operator(self, x)

    # -- for the plan cache -----------------------------------------------------------------------
    def factor_blocks(self):
        """(nterms, xfac or None, yfac) as contiguous [nterms][3][g] arrays for the C-ABI."""
        yfac = np.ascontiguousarray(np.stack([y for _, y in self.terms]))
        xfac = None if self.dimension == "1d" else np.ascontiguousarray(np.stack([x for x, _ in self.terms]))
        return len(self.terms), xfac, yfac

    def fingerprint(self):
        if self._fingerprint is None:
            h = hashlib.sha1()
            h.update(("%s:%d:%d" % (self.dimension, self.g, len(self.terms))).encode())
            for x, y in self.terms:
                if x is not None:
                    h.update(x.tobytes())
                h.update(y.tobytes())
            self._fingerprint = h.hexdigest()
        return self._fingerprint


def laplacian_operator(n, dimension="1d"):
    """Matrix-free counterpart of MGCMTStencilMaker.laplacian (MGCMTStencilMaker.py:15-25)."""
    n = int(n)
    if dimension == "1d":
        return StructuredOperator("1d", n, [(None, tri_laplacian(n))])
    # kronsum(L, L) = I (x) L + L (x) I   (MGCMTStencilMaker.py:23-24)
    return StructuredOperator("2d", n, [(tri_identity(n), tri_laplacian(n)), (tri_laplacian(n), tri_identity(n))])


def mehrstellen_operator(n):
    """The compact fourth-order ("Mehrstellen") 9-point Laplacian the reference's report proposes as the next fine-grid
    stencil (SURVEY.md par. 8(f)3; not in the reference's code):  1/(6 h^2) [[1, 4, 1], [4, -20, 4], [1, 4, 1]]
    =  I (x) L + L (x) I + (h^2 / 6) L (x) L  with the 1-D operator L of MGCMTStencilMaker.py:17-21 (h = 1/n, as there).
    Three Toeplitz terms: the fused kernels take it as a constant 9-point operator on every level."""
    n = int(n)
    h = 1. / n
    L = tri_laplacian(n)
    return StructuredOperator("2d", n, [(tri_identity(n), L.copy()), (L.copy(), tri_identity(n)), (L * (h ** 2 / 6.0), L.copy())])


def mehrstellen_mass(n):
    """M = I + (h^2 / 12) (I (x) L + L (x) I): the right-hand-side operator that goes with mehrstellen_operator —
    Delta_9 u = M (Delta u) + O(h^4), so  -Delta u = f  becomes  Delta_9 u = -M f, and  -Delta u = lambda u  the
    generalised problem  -Delta_9 u = lambda M u."""
    n = int(n)
    h = 1. / n
    L = tri_laplacian(n) * (h ** 2 / 12.0)
    y = L.copy()
    y[1] += 1.0
    return StructuredOperator("2d", n, [(tri_identity(n), y), (L.copy(), tri_identity(n))])


def identity_operator(n, dimension="1d"):
    """sparse.eye(N) as a structured operator (the mass matrix M of rqmin, RQMin.py:18)."""
    n = int(n)
    if dimension == "1d":
        return StructuredOperator("1d", n, [(None, tri_identity(n))])
    return StructuredOperator("2d", n, [(tri_identity(n), tri_identity(n))])


def potential_well_operator(g, depth, inner, scale=-1.0 / np.pi ** 2):
    """H = scale * laplacian(g, "2d") + diag(V) with the square-well potential of PotWellSolver.py:150-153 carried to
    2-D: V = `depth` outside the square [inner[0], inner[1])^2 of grid indices and 0 inside (BASELINE config 5).

    V = depth * (1 - chi (x) chi) is a sum of Kronecker products of diagonal factors, so H has three terms:
    I (x) (scale L + depth I)  +  (scale L) (x) I  -  (depth chi) (x) chi.
    """
    g = int(g)
    lo, hi = int(inner[0]), int(inner[1])
    chi = np.zeros(g)
    chi[lo:hi] = 1.0
    L = tri_laplacian(g) * float(scale)
    y1 = L.copy()
    y1[1] += float(depth)
    dchi = np.zeros((3, g))
    dchi[1] = chi
    return StructuredOperator("2d", g, [(tri_identity(g), y1), (L.copy(), tri_identity(g)), (dchi * (-float(depth)), dchi.copy())])


class UnrecognisedOperator(ValueError):
    pass


_CACHE = {}


def _digest(A):
    """Content digest of a scipy.sparse matrix (O(nnz), the cost of the structural check itself): a matrix mutated in
    place (``A *= c``, ``A.data[:] = ...``, ``setdiag``) keeps its id, shape and nnz but not this."""
    import hashlib
    h = hashlib.blake2b(digest_size=16)
    fmt = getattr(A, "format", None)
    h.update(str(fmt).encode())
    if fmt in ("csr", "csc", "bsr"):
        parts = (A.data, A.indices, A.indptr)
    elif fmt == "coo":
        parts = (A.data, A.row, A.col)
    elif fmt == "dia":
        parts = (A.data, A.offsets)
    else:                                                    # lil / dok / anything else: go through CSR
        B = A.tocsr()
        parts = (B.data, B.indices, B.indptr)
    for a in parts:
        h.update(np.ascontiguousarray(a).view(np.uint8).data)
    return h.digest()


def _cache_key(A):
    return (id(A), A.shape, getattr(A, "nnz", None), _digest(A))


def recognise(A, dimension=None):
    """StructuredOperator for a scipy.sparse matrix of the shapes the reference's callers build.

    1-D: any tridiagonal matrix.  2-D: a 5-point operator  I (x) Y + X (x) I  whose diagonal is
    d(i,j) = a(i) + b(j) [+ p(i) q(j)] — scaled / shifted Laplacians, separable potentials, and one product
    potential on top (a square well).  Anything else raises UnrecognisedOperator: the HIP path has no general-sparse kernels and there is no CPU
    fallback.
    """
    if isinstance(A, StructuredOperator):
        if dimension is not None and A.dimension != dimension:
            raise UnrecognisedOperator("operator is %s but dimension=%r was requested" % (A.dimension, dimension))
        return A
    if not sp.issparse(A):
        A = sp.csr_matrix(np.asarray(A, dtype=np.float64))
    tagged = getattr(A, "_mgcmt_structured", None)
    if tagged is not None and tagged[1] == _digest(A) and (dimension is None or tagged[0].dimension == dimension):
        return tagged[0]                      # a matrix this package assembled itself (e.g. a Galerkin level handed to a smoother)
    key = _cache_key(A)
    hit = _CACHE.get(key)
    if hit is not None and hit[0] is A and (dimension is None or hit[1].dimension == dimension):
        return hit[1]
    n = A.shape[0]
    if A.shape[0] != A.shape[1]:
        raise UnrecognisedOperator("operator must be square")
    if np.iscomplexobj(A):
        raise UnrecognisedOperator("complex operators are not supported by the HIP path")
    M = sp.csr_matrix(A, dtype=np.float64, copy=True)
    M.eliminate_zeros()
    nnz = M.nnz
    op = None
    if dimension in (None, "1d"):
        d0, dm, dp = M.diagonal(0), M.diagonal(-1), M.diagonal(1)
        if np.count_nonzero(d0) + np.count_nonzero(dm) + np.count_nonzero(dp) == nnz:
            y = np.zeros((3, n))
            y[0, 1:], y[1], y[2, :-1] = dm, d0, dp
            op = StructuredOperator("1d", n, [(None, y)])
    if op is None and dimension in (None, "2d"):
        g = int(round(math.sqrt(n)))
        if g * g != n:
            raise UnrecognisedOperator("2-D operator size %d is not a square number" % n)
        d0 = M.diagonal(0).reshape(g, g)
        e = np.zeros(n)
        e[:-1] = M.diagonal(1)
        w = np.zeros(n)
        w[1:] = M.diagonal(-1)
        s = np.zeros(n)
        s[:-g] = M.diagonal(g)
        nn = np.zeros(n)
        nn[g:] = M.diagonal(-g)
        counted = sum(np.count_nonzero(a) for a in (d0, e, w, s, nn))
        e, w, s, nn = e.reshape(g, g), w.reshape(g, g), s.reshape(g, g), nn.reshape(g, g)
        ok = counted == nnz and not e[:, -1].any() and not w[:, 0].any()
        ok = ok and np.array_equal(e, np.broadcast_to(e[0], (g, g))) and np.array_equal(w, np.broadcast_to(w[0], (g, g)))
        ok = ok and np.array_equal(s, np.broadcast_to(s[:, :1], (g, g))) and np.array_equal(nn, np.broadcast_to(nn[:, :1], (g, g)))
        extra = None
        if ok:
            c = 0.5 * d0[0, 0]
            yd = d0[0, :] - c
            xd = d0[:, 0] - d0[0, 0] + c
            scale = max(np.abs(d0).max(), 1e-300)
            rest = d0 - (xd[:, None] + yd[None, :])          # what an additively separable diagonal leaves over
            if np.abs(rest).max() > 4 * np.finfo(float).eps * scale:
                # a product potential on top, e.g. the square well V0 (1 - chi (x) chi) of PotWellSolver.py:150-153 in
                # 2-D: the remainder must be ONE outer product p (x) q (the kernels take three Kronecker terms)
                i0, j0 = np.unravel_index(np.argmax(np.abs(rest)), rest.shape)
                p_, q_ = rest[:, j0] / rest[i0, j0], rest[i0, :].copy()
                ok = np.abs(np.outer(p_, q_) - rest).max() <= 64 * np.finfo(float).eps * scale
                extra = (p_, q_)
        if not ok:
            op = _constant_nine_point(M, g)
            if op is None:
                raise UnrecognisedOperator(
                    "2-D operator is neither of the form I (x) Y + X (x) I (+ one product potential p (x) q on the diagonal) "
                    "nor a constant 9-point stencil; the HIP path handles scaled/shifted Laplacians with separable or "
                    "square-well potentials and constant compact stencils only")
            if len(_CACHE) > 64:
                _CACHE.clear()
            _CACHE[key] = (A, op)
            return op
        Y = np.zeros((3, g))
        Y[0], Y[1], Y[2] = w[0], yd, e[0]
        X = np.zeros((3, g))
        X[0], X[1], X[2] = nn[:, 0], xd, s[:, 0]
        terms = [(tri_identity(g), Y), (X, tri_identity(g))]
        if extra is not None:
            dp, dq = np.zeros((3, g)), np.zeros((3, g))
            dp[1], dq[1] = extra
            terms.append((dp, dq))
        op = StructuredOperator("2d", g, terms)
    if op is None:
        raise UnrecognisedOperator("operator is not tridiagonal (1-D)")
    if len(_CACHE) > 64:
        _CACHE.clear()
    _CACHE[key] = (A, op)
    return op


def _toeplitz_tri(lo, di, up, g):
    t = np.zeros((3, g))
    t[0, 1:], t[1], t[2, :-1] = lo, di, up
    return t


def _constant_nine_point(M, g):
    """A constant 3 x 3 stencil c on the g x g grid (zero Dirichlet truncation) — e.g. the Mehrstellen Laplacian, shifted
    or scaled — as  sum_a E_a (x) T(c[a, :])  with E_a the unit sub-/main/super-diagonal: exact, no arithmetic on the
    entries (rows 0 and 2 of a stencil that is symmetric top to bottom share one term).  None if M is not of that form."""
    if g < 3:
        return None
    r0 = g + 1
    row = M.getrow(r0)
    c = np.zeros((3, 3))
    for col, val in zip(row.indices, row.data):
        a, b = col // g - 1 + 1, col % g - 1 + 1
        if not (0 <= a <= 2 and 0 <= b <= 2):
            return None
        c[a, b] = val
    if np.array_equal(c[0], c[2]):
        terms = [(_toeplitz_tri(1.0, 0.0, 1.0, g), _toeplitz_tri(*c[0], g)), (tri_identity(g), _toeplitz_tri(*c[1], g))]
    else:
        terms = [(_toeplitz_tri(1.0, 0.0, 0.0, g), _toeplitz_tri(*c[0], g)), (tri_identity(g), _toeplitz_tri(*c[1], g)),
                 (_toeplitz_tri(0.0, 0.0, 1.0, g), _toeplitz_tri(*c[2], g))]
    terms = [(x, y) for x, y in terms if y.any()]
    if not terms:
        return None
    B = sum(sp.kron(tri_to_sparse(x), tri_to_sparse(y), format="csr") for x, y in terms).tocsr()
    D = (M - B).tocsr()
    if D.nnz and np.abs(D.data).max() != 0.0:
        return None
    return StructuredOperator("2d", g, terms)


def tag_structured(matrix, op):
    """Remember on a scipy matrix assembled from `op` what it was assembled from (with a content digest, so a matrix
    changed afterwards is not mistaken for it): ``recognise`` then maps it back without the structural check, also for
    the 9-point Galerkin operators that check does not cover."""
    matrix._mgcmt_structured = (op, _digest(matrix))
    return matrix
