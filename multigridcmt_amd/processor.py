"""MGCMTProcessor — the reference's vector utilities (MGCMTProcessor.py:4-73) on the GPU.

Columns are uploaded once, orthonormalised by the batched dot / axpy kernels of
csrc/kernels_blas.hip (scalars never leave the device inside one call) and downloaded.
"""
import numpy as np

from ._lib import SLOT_V, SLOT_W
from .operators import StructuredOperator, tri_identity
from .plan import get_plan


def _vector_plan(n, k):
    """A single-level 1 x n plan used as a vector workspace (the operator is irrelevant)."""
    n2 = 2
    while n2 < n:
        n2 *= 2
    op = StructuredOperator("1d", n2, [(None, tri_identity(n2))])
    return get_plan(op, n2, nvec=max(k, 1)), n2


def _upload_columns(plan, n2, a, slot=SLOT_V):
    n, k = a.shape
    buf = np.zeros(n2)
    for j in range(k):
        buf[:n] = a[:, j]
        plan.upload(0, slot, j, buf)


def _download_columns(plan, n, k, slot=SLOT_V):
    out = np.zeros((n, k))
    for j in range(k):
        out[:, j] = plan.download(0, slot, j)[:n]
    return out


class MGCMTProcessor:
    def __init__(self):
        pass

    def projection(self, v, u):
        """MGCMTProcessor.py:10-20 — (<v,u>/<u,u>) u."""
        v = np.asarray(v, dtype=np.float64).reshape(-1)
        u = np.asarray(u, dtype=np.float64).reshape(-1)
        plan, n2 = _vector_plan(len(u), 2)
        _upload_columns(plan, n2, np.column_stack((v, u)))
        inner1 = plan.dot(0, (SLOT_V, 0), (SLOT_V, 1))
        inner2 = plan.dot(0, (SLOT_V, 1), (SLOT_V, 1))
        plan.scale(0, inner1 / inner2, (SLOT_V, 1))
        return plan.download(0, SLOT_V, 1)[:len(u)]

    def gramschmidt(self, vectors, modified=1):
        """MGCMTProcessor.py:22-50 — classical (modified=0) or modified Gram-Schmidt of the columns;
        real output (the reference writes into np.zeros arrays, :31-32)."""
        a = np.array(vectors, dtype=np.float64)
        n, k = a.shape
        out = np.zeros((n, k))
        if k > 32:
            raise ValueError("at most 32 columns per call")
        plan, n2 = _vector_plan(n, k)
        _upload_columns(plan, n2, a)
        plan.gramschmidt(0, SLOT_V, k, modified=1 if modified else 0)
        out[:, :] = _download_columns(plan, n, k)
        return out

    def normalize(self, vectors):
        """MGCMTProcessor.py:52-63 — every column divided by its 2-norm."""
        a = np.array(vectors, dtype=np.float64)
        n, k = a.shape
        if k > 32:
            raise ValueError("at most 32 columns per call")
        plan, n2 = _vector_plan(n, k)
        _upload_columns(plan, n2, a)
        plan.normalize(0, SLOT_V, k)
        return _download_columns(plan, n, k)

    def orthogonality_check(self, vectors):
        """MGCMTProcessor.py:65-73 — Gram matrix <v_i, v_j>."""
        a = np.array(vectors, dtype=np.float64)
        n, k = a.shape
        if k > 32:
            raise ValueError("at most 32 columns per call")
        plan, n2 = _vector_plan(n, k)
        _upload_columns(plan, n2, a)
        gram = np.zeros((k, k))
        for i in range(k):
            for j in range(k):
                gram[i, j] = plan.dot(0, (SLOT_V, i), (SLOT_V, j))
        return gram
