"""Sharded V-cycle: the finest levels as row strips across ranks, one process per GPU.

Rank r owns rows [r*g/P, (r+1)*g/P) of every strip level.  Between two fused passes the
ranks exchange MGCMT_HALO_ROWS halo rows of V (and once per level of F) with their chain neighbours through
``torch.distributed`` point-to-point calls — RCCL over xGMI on GPUs, gloo in the CPU tests — and the fused
kernels recompute the few overlap rows redundantly, so one exchange serves a whole pass (two sweeps and a
transfer).  Below the switch level every rank holds the WHOLE coarse problem: one all-gather assembles the
restricted residual on all ranks, each runs the rest of the cycle redundantly with ``mgcmt_vcycle`` (a captured
HIP graph, identical arithmetic everywhere) and takes its own rows (with halo rows) of the correction — no rank
waits for another one's coarse solve and nothing has to be sent back.  Weighted Jacobi and the
multicolour Gauss-Seidel are order-independent, so the sharded cycle computes what the single-GPU cycle
computes (tests/test_distributed.py); the lexicographic smoothers do not shard.

PyTorch is plumbing here: tensors are zero-copy views of the plan's device memory (mgcmt_vec_ptr).
"""
import ctypes
import os

import numpy as np

from . import _lib
from ._lib import GS_MC, HALO_ROWS, SLOT_F, SLOT_T, SLOT_V, WJACOBI
from .plan import Plan


def _log2(x):
    return int(x).bit_length() - 1


class _DevicePointer:
    """Minimal __cuda_array_interface__ holder so torch can wrap plan memory without copying."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 3,
                                         "strides": None}


class ShardedPlan:
    def __init__(self, op, lowest, rank, world, device=0, switch_grid=None, on_gpu=True):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        if on_gpu and not torch.cuda.is_available():
            raise _lib.MgcmtError(
                "PyTorch sees no GPU.  In a process that uses both, import torch BEFORE the first call into "
                "libmgcmt_hip.so: each brings a HIP runtime and only the first one loaded can open the device.")
        self.rank, self.world, self.on_gpu = rank, world, on_gpu
        self.device = device
        # gloo moves host memory only: with device-resident strips (rehearsing several ranks on one GPU, or a box
        # without RCCL) every message is staged through a host tensor.  RCCL (backend "nccl") sends device memory.
        self.stage_host = bool(on_gpu and dist.is_initialized() and dist.get_backend() == "gloo")
        if op.dimension != "2d":
            raise ValueError("only 2-D problems are sharded")
        g = op.g
        if world < 1 or world & (world - 1) or g % world:
            raise ValueError("world size must be a power of two dividing the grid")
        nlev_total = _log2(g // lowest) + 1
        if switch_grid is None:
            # below this grid the cycle is cheaper run whole on every rank than as strips with two more exchanges per
            # level: a 2048^2 cycle costs 0.2 ms on one GPU and its all-gather moves 32 MiB; one level up it would be
            # 0.34 ms and 128 MiB (~0.3 ms over xGMI), more than the two exchanges (~0.1 ms each) it saves
            switch_grid = max(2048, 64 * world)
        # strip levels: grids above the switch size; every strip must keep >= 2*HALO_ROWS rows and even bounds
        ls = 0
        while ls < nlev_total - 1 and (g >> ls) > switch_grid and ((g >> ls) // world) >= 4 * HALO_ROWS:
            ls += 1
        if ls == 0:
            raise ValueError("grid %d too small to shard over %d ranks (switch grid %d)" % (g, world, switch_grid))
        self.strip_levels = ls                      # levels 0..ls-1 are smoothed as strips
        rows = g // world
        self.row_begin, self.row_end = rank * rows, (rank + 1) * rows
        # the strip plan also holds level `ls` as strips: it is the buffer the last restriction writes into
        # and the first prolongation reads from
        self.plan = Plan(op, g >> ls, nvec=1, device=device, row_begin=self.row_begin, row_end=self.row_end,
                         strip_levels=ls + 1)
        self.g, self.switch = g, g >> ls
        # every rank continues from the switch level on the whole grid (redundantly)
        from .operators import StructuredOperator
        terms = []
        xf = self.plan.factors(ls, 0)
        yf = self.plan.factors(ls, 1)
        for m in range(xf.shape[0]):
            terms.append((xf[m].copy(), yf[m].copy()))
        self.coarse = Plan(StructuredOperator("2d", self.switch, terms), lowest, nvec=1, device=device)
        self._views = {}
        self._fine_rhs_halo_valid = False
        self.use_recompute = True
        self.recompute_min_points = 1 << 22          # per-rank level size from which recomputing beats storing

    # -- zero-copy tensor views of plan memory ------------------------------------------------------
    def _flat(self, plan, level, slot):
        """1-D tensor over vector 0 of (plan, level, slot) INCLUDING its halo rows."""
        key = (id(plan), level, slot, plan.vec_ptr(level, slot, 0))
        t = self._views.get(key)
        if t is None:
            rows, cols, _ = plan.shapes[level]
            ptr = plan.vec_ptr(level, slot, 0) - HALO_ROWS * cols * 8
            count = (rows + 2 * HALO_ROWS) * cols
            if self.on_gpu:
                t = self.torch.as_tensor(_DevicePointer(ptr, count), device="cuda:%d" % self.device)
            else:
                buf = (ctypes.c_double * count).from_address(ptr)
                t = self.torch.from_numpy(np.ctypeslib.as_array(buf))
            self._views[key] = t
        return t

    def rows_view(self, plan, level, slot, first_row, nrows):
        """Tensor view of rows [first_row, first_row+nrows) (local indices; negative = upper halo)."""
        _, cols, _ = plan.shapes[level]
        flat = self._flat(plan, level, slot)
        a = (first_row + HALO_ROWS) * cols
        return flat[a:a + nrows * cols]

    # -- communication ------------------------------------------------------------------------------
    def _run_p2p(self, sends, recvs):
        """One batch of point-to-point operations: sends / recvs are lists of (tensor view, peer rank)."""
        dist = self.dist
        if not sends and not recvs:
            return
        if self.stage_host:
            staged = [(view, self.torch.empty(view.shape, dtype=view.dtype, device="cpu")) for view, _ in recvs]
            ops = [dist.P2POp(dist.isend, view.cpu(), peer) for view, peer in sends]
            ops += [dist.P2POp(dist.irecv, host, peer) for (_, peer), (_, host) in zip(recvs, staged)]
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            for view, host in staged:
                view.copy_(host)
            return
        ops = [dist.P2POp(dist.isend, view, peer) for view, peer in sends] + [dist.P2POp(dist.irecv, view, peer) for view, peer in recvs]
        for w in dist.batch_isend_irecv(ops):
            w.wait()

    def exchange_halo(self, *pairs):
        """Fill the halo rows of every (level, slot) in `pairs` with the neighbours' boundary rows (chain
        topology), all in ONE batch of point-to-point operations."""
        H = HALO_ROWS
        sends, recvs = [], []
        up, down = self.rank - 1, self.rank + 1
        for level, slot in pairs:
            rows = self.plan.shapes[level][0]
            if up >= 0:
                sends.append((self.rows_view(self.plan, level, slot, 0, H), up))
                recvs.append((self.rows_view(self.plan, level, slot, -H, H), up))
            if down < self.world:
                sends.append((self.rows_view(self.plan, level, slot, rows - H, H), down))
                recvs.append((self.rows_view(self.plan, level, slot, rows, H), down))
        self._run_p2p(sends, recvs)

    def gather_all(self, level, slot, dst_slot):
        """Strips of (level, slot) -> the whole-grid level 0 of every rank's coarse plan (one all-gather)."""
        dist = self.dist
        rows = self.plan.shapes[level][0]
        mine = self.rows_view(self.plan, level, slot, 0, rows)
        parts = [self.rows_view(self.coarse, 0, dst_slot, r * rows, rows) for r in range(self.world)]
        if self.world == 1:
            parts[0].copy_(mine)
        elif self.stage_host:
            host = [self.torch.empty(mine.shape, dtype=mine.dtype, device="cpu") for _ in range(self.world)]
            dist.all_gather(host, mine.cpu())
            for view, h in zip(parts, host):
                view.copy_(h)
        else:
            dist.all_gather(parts, mine)

    def take_own_rows(self, level, slot, src_slot):
        """This rank's rows of the coarse plan's whole-grid vector -> (level, slot), including the halo rows that
        exist (a local copy)."""
        H = HALO_ROWS
        rows = self.plan.shapes[level][0]
        total = rows * self.world
        lo, hi = max(self.rank * rows - H, 0), min((self.rank + 1) * rows + H, total)
        self.rows_view(self.plan, level, slot, lo - self.rank * rows, hi - lo).copy_(self.rows_view(self.coarse, 0, src_slot, lo, hi - lo))

    # -- data ---------------------------------------------------------------------------------------
    def set_shift(self, mu):
        self.plan.set_shifts([float(mu)])
        self.coarse.set_shifts([float(mu)])

    def upload_local(self, slot, host_rows):
        """This rank's rows of a fine-level vector (host array of local_rows*g doubles)."""
        self.plan.upload(0, slot, 0, host_rows)
        if slot == SLOT_F:
            self._fine_rhs_halo_valid = False

    def download_local(self, slot):
        return self.plan.download(0, slot, 0)

    def sync(self):
        self.plan.sync()

    # -- the cycle ----------------------------------------------------------------------------------
    def _passes(self, level, kind, nu):
        cap = self.plan.fused_max_sweeps(level, kind)
        if cap < 1:
            raise _lib.MgcmtError("strip level %d is not covered by the fused kernels" % level)
        out, left = [], nu
        while left > 0:
            n = min(cap, left)
            out.append(n)
            left -= n
        return out

    def vcycle(self, nu1, nu2, kind, omega=1.0, nu_coarse=None):
        """One V(nu1,nu2) cycle on V, F of the fine level (sharded down to the switch grid)."""
        if kind not in (WJACOBI, GS_MC):
            raise ValueError("only weighted Jacobi and multicolour Gauss-Seidel shard; lexicographic sweeps are sequential")
        if nu1 < 1 or nu2 < 1:
            raise ValueError("the sharded cycle needs at least one pre- and one post-smoothing sweep")
        nu_coarse = nu1 if nu_coarse is None else nu_coarse
        P, ls = self.plan, self.strip_levels
        recompute, still_zero = [0] * ls, [False] * ls
        for l in range(ls):
            nu, nu_up = (nu1, nu2) if l == 0 else (nu_coarse, nu_coarse)
            passes = self._passes(l, kind, nu)
            first_up = self._passes(l, kind, nu_up)[0]
            for i, n in enumerate(passes):
                last = i == len(passes) - 1
                zero_in = i == 0 and l > 0                 # the coarse iterate starts at zero: nothing to read or exchange
                if i == 0 and l == 0:
                    # F of the fine level is constant between uploads: its halo rows travel once
                    if self._fine_rhs_halo_valid:
                        self.exchange_halo((0, SLOT_V))
                    else:
                        self.exchange_halo((0, SLOT_V), (0, SLOT_F))
                        self._fine_rhs_halo_valid = True
                elif i == 0:
                    self.exchange_halo((l, SLOT_F))
                else:
                    self.exchange_halo((l, SLOT_V))
                mode = 2 if last else 0
                # recompute instead of store (bandwidth-bound levels): the last down-leg pass writes only the
                # restricted residual, the first up-leg pass re-runs its sweeps from the untouched V
                if last and self.use_recompute and P.size(l) >= self.recompute_min_points and n <= P.fused_max_recompute(l, kind, first_up):
                    mode |= 8
                    recompute[l], still_zero[l] = n, zero_in
                P.fused_pass(l, kind, n, omega=omega, mode=mode | (4 if zero_in else 0))
        # the coarse problem: all-gather, the same sub-cycle on every rank, own rows of the correction with halo rows
        self.gather_all(ls, SLOT_F, SLOT_F)
        self.coarse.vcycle(nu_coarse, nu_coarse, kind, omega=omega, k=1, nu_coarse=nu_coarse, level=0, zero_start=True)
        self.take_own_rows(ls, SLOT_V, SLOT_V)
        for l in range(ls - 1, -1, -1):
            nu = nu2 if l == 0 else nu_coarse
            passes = self._passes(l, kind, nu)
            for i, n in enumerate(passes):
                need = []
                if i == 0 and l + 1 < ls:
                    need.append((l + 1, SLOT_V))           # the correction to interpolate
                if not (i == 0 and recompute[l]):
                    need.append((l, SLOT_V))               # (a recomputing pass reads the V whose halo rows are still valid)
                if need:
                    self.exchange_halo(*need)
                mode = (1 | (4 if still_zero[l] else 0) | (recompute[l] << 4)) if i == 0 else 0
                P.fused_pass(l, kind, n, omega=omega, mode=mode)

    def residual_norm(self):
        """|| F - (A - mu I) V ||_2 over all ranks."""
        P = self.plan
        self.exchange_halo((0, SLOT_V))
        P.apply(0, (SLOT_V, 0), (SLOT_T, 0), with_shift=True)
        P.axpy(0, -1.0, (SLOT_F, 0), (SLOT_T, 0))
        local = P.dot(0, (SLOT_T, 0), (SLOT_T, 0))
        t = self.torch.tensor([local], dtype=self.torch.float64,
                              device=("cuda:%d" % self.device) if self.on_gpu and not self.stage_host else "cpu")
        self.dist.all_reduce(t)
        return float(t.item()) ** 0.5

    def close(self):
        self._views.clear()
        self.plan.close()
        self.coarse.close()
