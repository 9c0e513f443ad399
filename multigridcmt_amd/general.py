"""General sparse, complex operators on the device: the path behind ``operators.recognise``'s failure
branch for 1-D cycles — the k.p Hamiltonians of ThesisProblem.py:38-40,80,101 (a complex 4n x 4n block matrix cycled
as one 1-D grid of length 4n, smoother=solver.gseidel, lowest_level=2**5).

``CsrPlan`` wraps ``mgcmt_csr_plan`` of include/mgcmt_hip.h: the CSR matrix is uploaded once, the Galerkin hierarchy
R*A*P is built on the GPU, and smoothing, residuals, transfers, the coarsest solve and whole V-cycles run there.
There is no CPU fallback: what these kernels cannot take (2-D cycles of unstructured operators, a coarsest level of more
than 64 unknowns) raises.
"""
import ctypes
from collections import OrderedDict
from ctypes import c_int, c_int32, c_int64, c_void_p

import numpy as np
import scipy.sparse as sp

from . import _lib
from ._lib import SLOT_F, SLOT_T, SLOT_V, check


def _digest(A):
    import hashlib
    h = hashlib.blake2b(digest_size=16)
    for a in (A.data, A.indices, A.indptr):
        h.update(np.ascontiguousarray(a).view(np.uint8).data)
    return h.digest()


class CsrPlan:
    def __init__(self, A, lowest, device=0):
        A = sp.csr_matrix(A)
        A.sort_indices()
        n = A.shape[0]
        if A.shape[0] != A.shape[1]:
            raise ValueError("operator must be square")
        self.n, self.lowest = n, int(lowest)
        self._h = c_void_p()
        indptr = np.ascontiguousarray(A.indptr, dtype=np.int64)
        indices = np.ascontiguousarray(A.indices, dtype=np.int32)
        values = np.ascontiguousarray(A.data.astype(np.complex128)).view(np.float64)
        check(_lib.lib().mgcmt_csr_plan_create(device, n, self.lowest, indptr.ctypes.data_as(ctypes.POINTER(c_int64)),
                                               indices.ctypes.data_as(ctypes.POINTER(c_int32)), _lib.as_dp(values),
                                               ctypes.byref(self._h)))
        k = c_int(0)
        check(_lib.lib().mgcmt_csr_num_levels(self._h, ctypes.byref(k)))
        self.num_levels = k.value

    def close(self):
        if self._h:
            _lib.lib().mgcmt_csr_plan_destroy(self._h)
            self._h = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def level_info(self, level):
        n, nnz, chunk = c_int64(0), c_int64(0), c_int32(0)
        check(_lib.lib().mgcmt_csr_level_info(self._h, level, ctypes.byref(n), ctypes.byref(nnz), ctypes.byref(chunk)))
        return n.value, nnz.value, chunk.value

    def matrix(self, level):
        """The level's operator (level > 0: the Galerkin product computed on the device) as a scipy CSR matrix."""
        n, nnz, _ = self.level_info(level)
        indptr, indices, values = np.zeros(n + 1, dtype=np.int64), np.zeros(max(nnz, 1), dtype=np.int32), np.zeros(2 * max(nnz, 1))
        check(_lib.lib().mgcmt_csr_get_matrix(self._h, level, indptr.ctypes.data_as(ctypes.POINTER(c_int64)),
                                              indices.ctypes.data_as(ctypes.POINTER(c_int32)), _lib.as_dp(values)))
        return sp.csr_matrix((values.view(np.complex128)[:nnz], indices[:nnz], indptr), shape=(n, n))

    def upload(self, level, slot, host):
        a = np.ascontiguousarray(np.asarray(host).reshape(-1), dtype=np.complex128)
        check(_lib.lib().mgcmt_csr_upload(self._h, level, slot, _lib.as_dp(a.view(np.float64)), a.size, None))

    def download(self, level, slot):
        n = self.level_info(level)[0]
        out = np.empty(n, dtype=np.complex128)
        check(_lib.lib().mgcmt_csr_download(self._h, level, slot, _lib.as_dp(out.view(np.float64)), n, None))
        return out

    def apply(self, level, src, dst, shift=0.0):
        check(_lib.lib().mgcmt_csr_apply(self._h, level, src, dst, ctypes.c_double(shift), None))

    def smooth(self, level, kind, nu, omega=1.0, shift=0.0):
        check(_lib.lib().mgcmt_csr_smooth(self._h, level, kind, int(nu), ctypes.c_double(omega), ctypes.c_double(shift), None))

    def vcycle(self, nu1, nu2, kind, omega=1.0, nu_coarse=4, shift=0.0):
        check(_lib.lib().mgcmt_csr_vcycle(self._h, int(nu1), int(nu2), int(nu_coarse), kind, ctypes.c_double(omega),
                                          ctypes.c_double(shift), None))


_PLANS = OrderedDict()
_MAX_PLANS = 4


def get_csr_plan(A, lowest):
    """A cached device hierarchy for (matrix content, coarsest size)."""
    A = sp.csr_matrix(A)
    A.sort_indices()
    key = (A.shape, A.nnz, _digest(A), int(lowest))
    plan = _PLANS.get(key)
    if plan is not None:
        _PLANS.move_to_end(key)
        return plan
    plan = CsrPlan(A, lowest)
    _PLANS[key] = plan
    while len(_PLANS) > _MAX_PLANS:
        _, old = _PLANS.popitem(last=False)
        old.close()
    return plan


def release_plans():
    while _PLANS:
        _, p = _PLANS.popitem()
        p.close()


def real_if_real(x, *inputs):
    """The reference's results are complex exactly when one of its inputs is (NumPy promotion)."""
    if any(np.iscomplexobj(a) or (sp.issparse(a) and np.iscomplexobj(a.data)) for a in inputs):
        return x
    return np.ascontiguousarray(x.real)
