"""MGCMTStencilMaker — same public interface as the reference's class (MGCMTStencilMaker.py:5-78).

``laplacian`` / ``interpolation`` / ``restriction`` return ``scipy.sparse`` matrices exactly as the
reference does, because its callers do algebra with them (``-1/pi**2 * laplacian``, ``P * vec``,
``eigsh(H)``: 2DPotMatrixVcycle.py:26-27,59,79).  The V-cycle itself never uses these matrices: the
solver recognises the operator's structure and runs matrix-free HIP kernels whose transfer stencils
are the ones documented here.  ``matrix_free=True`` (an addition) returns a StructuredOperator for
grids too large to assemble.
"""
import math

import numpy as np
import scipy.sparse as spsparse

from .operators import laplacian_operator, mehrstellen_mass, mehrstellen_operator


def _power_of_two(x):
    p = math.log(x) / math.log(2)
    return p, float(p).is_integer()


class MGCMTStencilMaker:
    def __init__(self):
        pass

    def laplacian(self, n, dimension="1d", matrix_free=False):
        """MGCMTStencilMaker.py:15-25: 1-D tridiag(1,-2,1)/h^2 (h = 1/n, n x n CSC); 2-D kronsum."""
        n = int(n)
        if matrix_free:
            return laplacian_operator(n, dimension)
        if dimension == "1d":
            h = 1. / n
            lap = spsparse.diags([1, -2, 1], [-1, 0, 1], shape=(n, n), format="csc")
            return lap * (1 / h ** 2)
        if dimension == "2d":
            one_d = self.laplacian(n, dimension="1d")
            return spsparse.kronsum(one_d, one_d)
        return None

    def mehrstellen(self, n, matrix_free=False):
        """Not in the reference: the fourth-order compact 9-point Laplacian its report names as the next stencil
        (operators.mehrstellen_operator), as a sparse matrix like laplacian(n, "2d") or matrix-free."""
        op = mehrstellen_operator(int(n))
        return op if matrix_free else op.tocsr().tocsc()

    def mehrstellen_mass(self, n, matrix_free=False):
        """The operator M on the right-hand side of the Mehrstellen discretisation (operators.mehrstellen_mass)."""
        op = mehrstellen_mass(int(n))
        return op if matrix_free else op.tocsr().tocsc()

    def interpolation(self, old_gridsize, new_gridsize, dimension="1d"):
        """MGCMTStencilMaker.py:27-54.  Ratio m = new/old = 2^p: column J is the hat of half-width m
        centred on fine index (J+1)m-1 with peak 1 (for m = 2: 1/2, 1, 1/2 on rows 2J, 2J+1, 2J+2);
        2-D is kron(S, S).  Sizes that are not powers of two print a message and return None."""
        new_gridsize = int(new_gridsize)
        if dimension == "2d":
            s = self.interpolation(old_gridsize, new_gridsize, dimension="1d")
            return None if s is None else spsparse.kron(s, s, format="csc")
        if dimension != "1d":
            return None
        p_old, old_ok = _power_of_two(old_gridsize)
        p_new, new_ok = _power_of_two(new_gridsize)
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
        d = np.arange(-(m - 1), m)
        rows = ((np.arange(old)[:, None] + 1) * m - 1 + d[None, :]).ravel()
        cols = np.repeat(np.arange(old), len(d))
        vals = np.tile((m - np.abs(d)) / float(m), old)
        keep = (rows >= 0) & (rows < new_gridsize)
        return spsparse.csc_matrix((vals[keep], (rows[keep], cols[keep])), shape=(new_gridsize, old))

    def restriction(self, old_gridsize, new_gridsize, dimension="1d"):
        """MGCMTStencilMaker.py:57-78.  1-D: (1/2)^p P^T — full weighting (1/4,1/2,1/4) on fine
        2J..2J+2 for one level.  2-D: 1/4 kron(S,S)^T; the 1/4 is NOT raised to the level difference
        (:77), so a two-level jump has row sums 4 — kept."""
        if dimension == "2d":
            p = self.interpolation(new_gridsize, old_gridsize, dimension="2d")
            return None if p is None else 1. / 4. * p.T
        if dimension != "1d":
            return None
        p_old, old_ok = _power_of_two(old_gridsize)
        p_new, new_ok = _power_of_two(new_gridsize)
        if not p_new < p_old:
            print("New gridsize is bigger (more elements) than old gridsize !")
            return None
        if not old_ok:
            print("Old gridsize isn't a power of 2 !")
            return None
        if not new_ok:
            print("New gridsize isn't a power of 2 !")
            return None
        p = self.interpolation(new_gridsize, old_gridsize)
        return spsparse.csc_matrix((1. / 2) ** (p_old - p_new) * p.T)
