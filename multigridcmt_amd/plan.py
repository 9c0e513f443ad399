"""Python handle on a device hierarchy (``mgcmt_plan`` of include/mgcmt_hip.h).

One plan = one operator (plus optional mass operator) on one grid size with a fixed coarsest level:
the Galerkin factors of every level (MGCMTSolver.py:318) and the level vectors live on the GPU.
"""
import ctypes
from collections import OrderedDict
from ctypes import c_double, c_int, c_int64, c_void_p

import numpy as np

from . import _lib, hostmem
from ._lib import OP_A, OP_M, SLOT_F, SLOT_T, SLOT_V, SLOT_W, PlanDesc, as_dp, check


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Plan:
    def __init__(self, op, lowest, nvec=1, mass=None, device=0, row_begin=0, row_end=0, strip_levels=0):
        self._h = c_void_p()
        self.op = op
        self.mass = mass
        self.dim = 1 if op.dimension == "1d" else 2
        self.g = op.g
        self.lowest = int(lowest)
        self.nvec = int(nvec)
        self.device = device
        nterms, xfac, yfac = op.factor_blocks()
        desc = PlanDesc()
        desc.dim, desc.nterms, desc.g, desc.lowest = self.dim, nterms, self.g, self.lowest
        desc.xfac = as_dp(xfac) if xfac is not None else None
        desc.yfac = as_dp(yfac)
        keep = [xfac, yfac]
        if mass is not None:
            mt, mx, my = mass.factor_blocks()
            desc.m_nterms = mt
            desc.m_xfac = as_dp(mx) if mx is not None else None
            desc.m_yfac = as_dp(my)
            keep += [mx, my]
        desc.nvec, desc.device = self.nvec, device
        desc.row_begin, desc.row_end, desc.strip_levels = row_begin, row_end, strip_levels
        check(_lib.lib().mgcmt_plan_create(ctypes.byref(desc), ctypes.byref(self._h)))
        del keep
        n = c_int(0)
        check(_lib.lib().mgcmt_plan_num_levels(self._h, ctypes.byref(n)))
        self.num_levels = n.value
        self.shapes = [self.level_shape(l) for l in range(self.num_levels)]
        self._shifts = None

    # -- lifetime ---------------------------------------------------------------------------------
    def close(self):
        if self._h:
            _lib.lib().mgcmt_plan_destroy(self._h)
            self._h = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- geometry ---------------------------------------------------------------------------------
    def level_shape(self, level):
        r, c, b = c_int64(0), c_int64(0), c_int64(0)
        check(_lib.lib().mgcmt_plan_level_shape(self._h, level, ctypes.byref(r), ctypes.byref(c), ctypes.byref(b)))
        return r.value, c.value, b.value

    def size(self, level=0):
        r, c, _ = self.shapes[level]
        return r * c

    def factors(self, level, which, op=OP_A):
        """Host copy of a level's Kronecker factors: array [nterms, 3, n]."""
        src = self.op if op == OP_A else self.mass
        nterms = len(src.terms)
        n = (self.g >> level) if (which == 1 or self.dim == 2) else 1
        out = np.zeros((nterms, 3, n))
        check(_lib.lib().mgcmt_plan_get_factors(self._h, op, level, which, as_dp(out), out.size))
        return out

    def level_halo(self, level):
        """(halo rows kept around every vector of `level`, how many of them a sharded cycle exchanges and reads)"""
        h, x = c_int(0), c_int(0)
        check(_lib.lib().mgcmt_plan_level_halo(self._h, level, ctypes.byref(h), ctypes.byref(x)))
        return h.value, x.value

    def vec_ptr(self, level, slot, vec=0):
        p = c_void_p()
        check(_lib.lib().mgcmt_vec_ptr(self._h, level, slot, vec, ctypes.byref(p)))
        return p.value

    # -- data -------------------------------------------------------------------------------------
    def set_shifts(self, shifts, stream=None):
        s = _f64(np.atleast_1d(shifts)).reshape(-1)
        check(_lib.lib().mgcmt_set_shifts(self._h, as_dp(s), len(s), stream))
        self._shifts = s.copy()

    def upload(self, level, slot, vec, host, stream=None):
        a = _f64(host).reshape(-1)
        check(_lib.lib().mgcmt_upload(self._h, level, slot, vec, as_dp(a), a.size, stream))

    def download(self, level, slot, vec, stream=None):
        out = hostmem.empty(self.size(level))  # (large results: recycled page-locked memory, one DMA)
        check(_lib.lib().mgcmt_download(self._h, level, slot, vec, as_dp(out), out.size, stream))
        return out

    def download_into(self, level, slot, vec, out, stream=None):
        """Like download, into a caller-provided contiguous float64 array of the level's size."""
        if not (out.flags.c_contiguous and out.dtype == np.float64 and out.size == self.size(level)):
            raise ValueError("download_into needs a contiguous float64 array of the level's size")
        check(_lib.lib().mgcmt_download(self._h, level, slot, vec, as_dp(out), out.size, stream))

    def fill(self, level, slot, vec, value, stream=None):
        check(_lib.lib().mgcmt_fill(self._h, level, slot, vec, c_double(value), stream))

    def zero(self, level, slot, vec, stream=None):
        check(_lib.lib().mgcmt_zero(self._h, level, slot, vec, stream))

    def copy(self, level, src_slot, src_vec, dst_slot, dst_vec, stream=None):
        check(_lib.lib().mgcmt_copy(self._h, level, src_slot, src_vec, dst_slot, dst_vec, stream))

    def sync(self, stream=None):
        check(_lib.lib().mgcmt_sync(stream))

    # -- the path ---------------------------------------------------------------------------------
    def smooth(self, level, kind, nu, omega=1.0, k=1, stream=None):
        check(_lib.lib().mgcmt_smooth(self._h, level, kind, nu, c_double(omega), k, stream))

    def residual_restrict(self, level, k=1, stream=None):
        check(_lib.lib().mgcmt_residual_restrict(self._h, level, k, stream))

    def prolong_correct(self, level, k=1, stream=None):
        check(_lib.lib().mgcmt_prolong_correct(self._h, level, k, stream))

    def coarse_solve(self, level, k=1, stream=None):
        check(_lib.lib().mgcmt_coarse_solve(self._h, level, k, stream))

    def vcycle(self, nu1, nu2, kind, omega=1.0, k=1, nu_coarse=4, gram_schmidt=False, level=0, stream=None, zero_start=False):
        """zero_start: the iterate on `level` is to be taken as zero, whatever V holds (the reference's eigen-drivers
        start every cycle this way, 1DPotMatrixVcycle.py:70).  The library's first pass then neither reads V nor
        needs it cleared (MGCMT_CYCLE_ZERO_START)."""
        flags = (_lib.CYCLE_GRAM_SCHMIDT if gram_schmidt else 0) | (_lib.CYCLE_ZERO_START if zero_start else 0)
        check(_lib.lib().mgcmt_vcycle(self._h, level, nu1, nu2, nu_coarse, kind, c_double(omega), k, flags, stream))

    def twogrid(self, nu1, nu2, kind, omega=1.0, k=1, level=0, stream=None):
        check(_lib.lib().mgcmt_twogrid(self._h, level, nu1, nu2, kind, c_double(omega), k, stream))

    def apply(self, level, src, dst, op=OP_A, with_shift=False, stream=None):
        check(_lib.lib().mgcmt_apply(self._h, op, level, src[0], src[1], dst[0], dst[1], 1 if with_shift else 0, stream))

    def restrict(self, level, src, dst, stream=None):
        check(_lib.lib().mgcmt_restrict(self._h, level, src[0], src[1], dst[0], dst[1], stream))

    def prolong(self, level, src, dst, accumulate=False, stream=None):
        check(_lib.lib().mgcmt_prolong(self._h, level, src[0], src[1], dst[0], dst[1], 1 if accumulate else 0, stream))

    def dot(self, level, a, b, stream=None):
        out = c_double(0.0)
        check(_lib.lib().mgcmt_dot(self._h, level, a[0], a[1], b[0], b[1], ctypes.byref(out), stream))
        return out.value

    def gram(self, level, vectors, stream=None):
        """Gram matrix (len(vectors) <= 6) of the (slot, vec) pairs in one pass; synchronises."""
        nv = len(vectors)
        slots = (ctypes.c_int * nv)(*[v[0] for v in vectors])
        vecs = (ctypes.c_int * nv)(*[v[1] for v in vectors])
        out = np.zeros((nv, nv))
        check(_lib.lib().mgcmt_gram(self._h, level, nv, slots, vecs, _lib.as_dp(out), stream))
        return out

    def ritz_pair(self, level, x, w, scratch, stream=None):
        """(<x,x>, <x,w>, <w,w>, <x,A w>, <w,A w>) with the unshifted operator of `level`: the 2 x 2 Rayleigh-Ritz problem
        on span{x, w} when <x, A x> is known.  One pass over x and w on 2-D 5-point levels (nothing stored); `scratch`
        ((slot, vec), distinct from x and w) receives A w elsewhere.  Synchronises."""
        out = np.zeros(5)
        check(_lib.lib().mgcmt_ritz_pair(self._h, level, x[0], x[1], w[0], w[1], scratch[0], scratch[1], as_dp(out), stream))
        return out

    def rayleigh_residual(self, level, slot, k, stream=None):
        """(rq, res): Rayleigh quotients <v, A v>/<v, v> and residual norms ||(A - mu I) v|| of columns 0..k-1 of `slot`
        (mu = the plan's shifts); one synchronisation for all columns.  Slot W is the scratch."""
        rq, res = np.zeros(k), np.zeros(k)
        check(_lib.lib().mgcmt_rayleigh_residual(self._h, level, slot, k, as_dp(rq), as_dp(res), stream))
        return rq, res

    def rqmin(self, level, slot, vecs, nu, robust=False, want_rho=True, stream=None):
        """nu steps of the reference's rqmin (MGCMTSolver.py:17-57) on `level`, resident on the device (mgcmt_rqmin): vecs[0]
        of `slot` is the iterate, vecs[1..5] work space.  Returns rho (one synchronisation) or None (want_rho=False: none)."""
        v = (ctypes.c_int * 6)(*[int(i) for i in vecs])
        rho = c_double(0.0)
        check(_lib.lib().mgcmt_rqmin(self._h, level, slot, v, int(nu), 1 if robust else 0, ctypes.byref(rho) if want_rho else None, stream))
        return rho.value if want_rho else None

    def vcycle_rqmg(self, slot, vecs, nu1, nu2, robust=False, want_rho=True, stream=None):
        """One cycle of the reference's Rayleigh-quotient multigrid (MGCMTSolver.py:99-122) over all levels of the plan,
        resident on the device (mgcmt_vcycle_rqmg); vecs as for rqmin, on every level."""
        v = (ctypes.c_int * 6)(*[int(i) for i in vecs])
        rho = c_double(0.0)
        check(_lib.lib().mgcmt_vcycle_rqmg(self._h, slot, v, int(nu1), int(nu2), 1 if robust else 0, ctypes.byref(rho) if want_rho else None, stream))
        return rho.value if want_rho else None

    def rq_line_step(self, level, x, w, x_out, g, work=None, robust=False, record=-1, stream=None):
        """x_out = x + delta w minimising the Rayleigh quotient over span{x, w} (the 2 x 2 problem of rqmin,
        MGCMTSolver.py:33-50) and g = 2 (A x_out - rho M x_out), on the device without a host round trip
        (mgcmt_rq_line_step); vectors are (slot, vec) pairs.  w=None: rho and g of x alone.  record >= 0: rho is kept as
        that entry of the device-side history (rq_history)."""
        def pair(v):
            return None if v is None else (ctypes.c_int * 2)(int(v[0]), int(v[1]))
        check(_lib.lib().mgcmt_rq_line_step(self._h, level, pair(x), pair(w), pair(x_out), pair(g), pair(work), 1 if robust else 0, int(record),
                                            stream))

    def rq_history(self, first, count, stream=None):
        """entries [first, first + count) of the Rayleigh quotients recorded by rq_line_step (one synchronisation)"""
        out = np.empty(int(count))
        check(_lib.lib().mgcmt_rq_history(self._h, int(first), int(count), _lib.as_dp(out), stream))
        return out

    def lincomb(self, level, terms, dst, stream=None):
        """dst <- sum of coeff * (slot, vec) over `terms` = [(coeff, (slot, vec)), ...] (at most four)."""
        nt = len(terms)
        coeffs = np.array([float(c) for c, _ in terms])
        slots = (ctypes.c_int * nt)(*[v[0] for _, v in terms])
        vecs = (ctypes.c_int * nt)(*[v[1] for _, v in terms])
        check(_lib.lib().mgcmt_lincomb(self._h, level, nt, _lib.as_dp(coeffs), slots, vecs, dst[0], dst[1], stream))

    def block_gram(self, level, a, b, stream=None):
        """A^T B for lists of (slot, vec) pairs, len(a) <= 12, len(b) <= 4, in one pass; synchronises."""
        na, nb = len(a), len(b)
        out = np.zeros((na, nb))
        check(_lib.lib().mgcmt_block_gram(self._h, level, na, (ctypes.c_int * na)(*[v[0] for v in a]), (ctypes.c_int * na)(*[v[1] for v in a]),
                                          nb, (ctypes.c_int * nb)(*[v[0] for v in b]), (ctypes.c_int * nb)(*[v[1] for v in b]),
                                          _lib.as_dp(out), stream))
        return out

    def block_combine(self, level, inputs, outputs, coeffs, stream=None):
        """outputs[j] <- sum_i coeffs[i, j] * inputs[i] ((slot, vec) pairs; <= 12 inputs, <= 4 distinct outputs; an output
        may be one of the inputs)."""
        nin, nout = len(inputs), len(outputs)
        c = np.ascontiguousarray(np.asarray(coeffs, dtype=np.float64).reshape(nin, nout))
        check(_lib.lib().mgcmt_block_combine(self._h, level, nin, (ctypes.c_int * nin)(*[v[0] for v in inputs]),
                                             (ctypes.c_int * nin)(*[v[1] for v in inputs]), nout,
                                             (ctypes.c_int * nout)(*[v[0] for v in outputs]), (ctypes.c_int * nout)(*[v[1] for v in outputs]),
                                             _lib.as_dp(c), stream))

    def axpy(self, level, alpha, x, y, stream=None):
        check(_lib.lib().mgcmt_axpy(self._h, level, c_double(alpha), x[0], x[1], y[0], y[1], stream))

    def scale(self, level, alpha, v, stream=None):
        check(_lib.lib().mgcmt_scale(self._h, level, c_double(alpha), v[0], v[1], stream))

    def gramschmidt(self, level, slot, k, modified=1, stream=None):
        check(_lib.lib().mgcmt_gramschmidt(self._h, level, slot, k, 1 if modified else 0, stream))

    def normalize(self, level, slot, k, stream=None):
        check(_lib.lib().mgcmt_normalize(self._h, level, slot, k, stream))

    def fused_pass(self, level, kind, nsweep, omega=1.0, mode=0, k=1, stream=None):
        check(_lib.lib().mgcmt_fused_pass(self._h, level, kind, nsweep, c_double(omega), mode, k, stream))

    def fused_max_sweeps(self, level, kind):
        n = c_int(0)
        check(_lib.lib().mgcmt_fused_max_sweeps(self._h, level, kind, ctypes.byref(n)))
        return n.value

    def operator_kind(self, level):
        """One of _lib.OPK_*: how the library recognised the operator of `level` (decides the kernels it runs on)."""
        n = ctypes.c_int(0)
        check(_lib.lib().mgcmt_level_operator_kind(self._h, level, ctypes.byref(n)))
        return n.value

    def fused_max_recompute(self, level, kind, nsweep):
        n = c_int(0)
        check(_lib.lib().mgcmt_fused_max_recompute(self._h, level, kind, nsweep, ctypes.byref(n)))
        return n.value

    def set_option(self, option, value):
        check(_lib.lib().mgcmt_plan_set_option(self._h, option, int(value)))

    def bandwidth_probe(self, level, kind, blocks, reps, stream=None):
        ms = c_double(0.0)
        check(_lib.lib().mgcmt_bandwidth_probe(self._h, level, kind, blocks, reps, ctypes.byref(ms), stream))
        return ms.value

    def time_smoother(self, level, kind, nu, omega, reps, stream=None):
        ms = c_double(0.0)
        check(_lib.lib().mgcmt_time_smoother(self._h, level, kind, nu, c_double(omega), reps, ctypes.byref(ms), stream))
        return ms.value

    def time_fused_pass(self, level, kind, nsweep, omega, mode, reps, stream=None):
        """average milliseconds of one fused pass (mode as fused_pass) over `reps` launches, HIP events on the stream"""
        ms = c_double(0.0)
        check(_lib.lib().mgcmt_time_fused_pass(self._h, level, kind, nsweep, c_double(omega), mode, reps, ctypes.byref(ms), stream))
        return ms.value


# --------------------------------------------------------------------------------------------------
# plan cache: callers of the reference API pass the operator on every call
# --------------------------------------------------------------------------------------------------

_PLANS = OrderedDict()
_MAX_PLANS = 6


def _log2(x):
    return int(x).bit_length() - 1


def get_plan(op, lowest, nvec=1, mass=None):
    """A cached plan for (operator, coarsest size, mass operator) with room for at least nvec vectors."""
    key = (op.fingerprint(), int(lowest), None if mass is None else mass.fingerprint())
    plan = _PLANS.get(key)
    if plan is not None and plan.nvec >= nvec:
        _PLANS.move_to_end(key)
        return plan
    if plan is not None:
        plan.close()
        del _PLANS[key]
    plan = Plan(op, lowest, nvec=nvec, mass=mass)
    _PLANS[key] = plan
    while len(_PLANS) > _MAX_PLANS:
        _, old = _PLANS.popitem(last=False)
        old.close()
    return plan


def release_plans():
    """Free every cached device hierarchy."""
    while _PLANS:
        _, p = _PLANS.popitem()
        p.close()


def apply_operator(op, x):
    """A @ x on the GPU for a StructuredOperator (x: (n,), (n,1) or (n,k))."""
    x = np.asarray(x, dtype=np.float64)
    n = op.shape[0]
    cols = x.reshape(n, -1)
    plan = get_plan(op, op.g, nvec=1)      # a single level is enough
    out = np.empty_like(cols)
    for c in range(cols.shape[1]):
        plan.upload(0, SLOT_V, 0, cols[:, c])
        plan.apply(0, (SLOT_V, 0), (SLOT_T, 0))
        out[:, c] = plan.download(0, SLOT_T, 0)
    return out.reshape(x.shape)
