"""Sharded V-cycle: the finest levels as row strips across ranks, one process per GPU.

Rank r owns rows [r*g/P, (r+1)*g/P) of every strip level.  The cycle itself — which pass runs when,
what is exchanged after it, the all-gather in front of the coarse problem — is ``mgcmt_sharded_vcycle`` of
libmgcmt_hip.so (csrc/sharded.hip); this module only builds the two plans (strips + the whole coarse grid that every
rank solves redundantly), hands the library a transport and keeps track of which halo rows are still valid.

Transports
* ``"rccl"``  — the library calls RCCL itself (ncclSend/ncclRecv groups, ncclAllGather on HIP streams; no Python, no
  host synchronisation inside a cycle; the exchange of a pass's boundary rows overlaps its interior launch).  The
  128-byte communicator id comes from rank 0 (``rccl_unique_id``) through any channel the host program has — bench.py
  uses the torch.distributed store.
* ``"torch"`` — callbacks into ``torch.distributed`` point-to-point / all-gather (gloo in the CPU tests, where "device"
  memory is host memory of the emulated kernels; gloo with staging through host memory for several ranks on one GPU).

Weighted Jacobi and the multicolour Gauss-Seidel are order-independent, so the sharded cycle computes what the
single-GPU cycle computes (tests/test_distributed.py); the lexicographic smoothers do not shard.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import GS_MC, HALO_ROWS, SLOT_F, SLOT_T, SLOT_V, WJACOBI, check
from .plan import Plan


def _log2(x):
    return int(x).bit_length() - 1


def rccl_unique_id():
    """128 bytes naming a new RCCL communicator (call on ONE rank, distribute to all)."""
    buf = ctypes.create_string_buffer(_lib.UNIQUE_ID_BYTES)
    check(_lib.lib().mgcmt_comm_unique_id(buf))
    return buf.raw


class _DevicePointer:
    """Minimal __cuda_array_interface__ holder so torch can wrap plan memory without copying."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 3,
                                         "strides": None}


class _TorchTransport:
    """The external-transport callbacks of mgcmt_comm_init_external on top of torch.distributed."""

    def __init__(self, device, on_gpu):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.device, self.on_gpu = torch, dist, device, on_gpu
        # gloo moves host memory only: device-resident strips are staged through host tensors
        self.stage_host = bool(on_gpu and dist.is_initialized() and dist.get_backend() == "gloo")
        self.error = None
        self.p2p = _lib.P2P_FN(self._p2p)
        self.allgather = _lib.ALLGATHER_FN(self._allgather)
        self.allreduce = _lib.ALLREDUCE_FN(self._allreduce)

    def _tensor(self, ptr, count):
        if self.on_gpu:
            return self.torch.as_tensor(_DevicePointer(ptr, count), device="cuda:%d" % self.device)
        buf = (ctypes.c_double * count).from_address(ptr)
        return self.torch.from_numpy(np.ctypeslib.as_array(buf))

    def _p2p(self, user, nops, ops):
        try:
            dist, torch = self.dist, self.torch
            sends = [(self._tensor(ops[i].ptr, ops[i].count), ops[i].peer) for i in range(nops) if ops[i].is_send]
            recvs = [(self._tensor(ops[i].ptr, ops[i].count), ops[i].peer) for i in range(nops) if not ops[i].is_send]
            if self.stage_host:
                staged = [torch.empty(v.shape, dtype=v.dtype, device="cpu") for v, _ in recvs]
                batch = [dist.P2POp(dist.isend, v.cpu(), peer) for v, peer in sends]
                batch += [dist.P2POp(dist.irecv, h, peer) for (_, peer), h in zip(recvs, staged)]
                for w in dist.batch_isend_irecv(batch):
                    w.wait()
                for (v, _), h in zip(recvs, staged):
                    v.copy_(h)
                if self.on_gpu:
                    torch.cuda.synchronize()
                return 0
            batch = [dist.P2POp(dist.isend, v, peer) for v, peer in sends] + [dist.P2POp(dist.irecv, v, peer) for v, peer in recvs]
            for w in dist.batch_isend_irecv(batch):
                w.wait()
            if self.on_gpu:
                torch.cuda.synchronize()
            return 0
        except Exception as e:                                   # an exception must not unwind through the C frames
            self.error = e
            return -1

    def _allgather(self, user, send, recv, count):
        try:
            dist, torch = self.dist, self.torch
            world = dist.get_world_size()
            mine = self._tensor(send, count)
            whole = self._tensor(recv, count * world)
            parts = [whole[r * count:(r + 1) * count] for r in range(world)]
            if self.stage_host:
                host = [torch.empty(mine.shape, dtype=mine.dtype, device="cpu") for _ in range(world)]
                dist.all_gather(host, mine.cpu())
                for view, h in zip(parts, host):
                    view.copy_(h)
            else:
                dist.all_gather(parts, mine)
            if self.on_gpu:
                torch.cuda.synchronize()
            return 0
        except Exception as e:
            self.error = e
            return -1

    def _allreduce(self, user, inout, n):
        try:
            arr = np.ctypeslib.as_array(inout, shape=(n,))
            dev = ("cuda:%d" % self.device) if self.on_gpu and not self.stage_host else "cpu"
            t = self.torch.tensor(arr.copy(), dtype=self.torch.float64, device=dev)
            self.dist.all_reduce(t)
            arr[:] = t.cpu().numpy()
            return 0
        except Exception as e:
            self.error = e
            return -1


class ShardedPlan:
    """nvec: columns held (vcycle_matrix's k, each with its own shift).  emulate=(R, N): a ONE-rank communicator whose
    strip plan is rank R's share of an N-rank job — the cycle does that rank's work with itself as both neighbours
    (MGCMT_COMM_OPT_EMULATE_OF): a timing rehearsal of one rank of N on one GPU, not the N-rank job's numbers."""

    def __init__(self, op, lowest, rank, world, device=0, switch_grid=None, on_gpu=True, transport="torch", unique_id=None, nvec=1,
                 emulate=None):
        self.rank, self.world, self.on_gpu, self.device = rank, world, on_gpu, device
        self.nvec = int(nvec)
        if emulate is not None:
            if world != 1:
                raise ValueError("rank emulation runs on a one-rank communicator")
            erank, world_e = int(emulate[0]), int(emulate[1])
        else:
            erank, world_e = rank, world
        if op.dimension != "2d":
            raise ValueError("only 2-D problems are sharded")
        g = op.g
        if world_e < 1 or world_e & (world_e - 1) or g % world_e or not 0 <= erank < world_e:
            raise ValueError("world size must be a power of two dividing the grid")
        nlev_total = _log2(g // lowest) + 1
        if switch_grid is None:
            # below this grid the cycle is cheaper run whole on every rank than as strips with two more exchanges per
            # level: a 2048^2 cycle costs 0.2 ms on one GPU and its all-gather moves 32 MiB; one level up it would be
            # 0.34 ms and 128 MiB (~0.3 ms over xGMI), more than the two exchanges it saves
            switch_grid = max(2048, 64 * world_e)
        # strip levels: grids above the switch size; every strip must keep >= 4*HALO_ROWS rows and even bounds
        ls = 0
        while ls < nlev_total - 1 and (g >> ls) > switch_grid and ((g >> ls) // world_e) >= 4 * HALO_ROWS:
            ls += 1
        if ls == 0:
            raise ValueError("grid %d too small to shard over %d ranks (switch grid %d)" % (g, world_e, switch_grid))
        self.strip_levels = ls                      # levels 0..ls-1 are smoothed as strips
        rows = g // world_e
        self.row_begin, self.row_end = erank * rows, (erank + 1) * rows
        # the strip plan also holds level `ls` as strips: it is the buffer the last restriction writes into
        # and the first prolongation reads from
        self.plan = Plan(op, g >> ls, nvec=self.nvec, device=device, row_begin=self.row_begin, row_end=self.row_end,
                         strip_levels=ls + 1)
        self.g, self.switch = g, g >> ls
        # every rank continues from the switch level on the whole grid (redundantly)
        from .operators import StructuredOperator
        xf, yf = self.plan.factors(ls, 0), self.plan.factors(ls, 1)
        terms = [(xf[m].copy(), yf[m].copy()) for m in range(xf.shape[0])]
        self.coarse = Plan(StructuredOperator("2d", self.switch, terms), lowest, nvec=self.nvec, device=device)
        self.transport = transport
        self._torch_transport = None
        L = _lib.lib()
        if transport == "rccl":
            if unique_id is None or len(unique_id) != _lib.UNIQUE_ID_BYTES:
                raise ValueError("the RCCL transport needs the %d-byte id of rccl_unique_id() from rank 0" % _lib.UNIQUE_ID_BYTES)
            check(L.mgcmt_comm_init(self.plan._h, rank, world, ctypes.create_string_buffer(unique_id, _lib.UNIQUE_ID_BYTES)))
        elif transport == "torch":
            import torch
            if on_gpu and not torch.cuda.is_available():
                raise _lib.MgcmtError(
                    "PyTorch sees no GPU.  In a process that uses both, import torch BEFORE the first call into "
                    "libmgcmt_hip.so: each brings a HIP runtime and only the first one loaded can open the device.")
            t = self._torch_transport = _TorchTransport(device, on_gpu)
            check(L.mgcmt_comm_init_external(self.plan._h, rank, world, t.p2p, t.allgather, t.allreduce, None))
        else:
            raise ValueError("transport must be 'rccl' or 'torch'")
        self._v_halo_valid = False
        self._f_halo_valid = False
        self.emulate = None
        if emulate is not None:
            self.emulate = (erank, world_e)
            self.set_comm_option(_lib.COMM_OPT_SELF_RING, 1)
            self.set_comm_option(_lib.COMM_OPT_EMULATE_OF, world_e)

    def _check(self, rc):
        t = self._torch_transport
        if rc != 0 and t is not None and t.error is not None:
            err, t.error = t.error, None
            raise err
        check(rc)

    def set_comm_option(self, option, value):
        check(_lib.lib().mgcmt_comm_set_option(self.plan._h, option, int(value)))

    # -- communication ------------------------------------------------------------------------------
    def exchange_halo(self, *pairs, ring=False, k=1):
        """Fill the halo rows of every (level, slot) in `pairs` with the neighbours' boundary rows (chain topology);
        one batch per level."""
        by_level = {}
        for level, slot in pairs:
            by_level[level] = by_level.get(level, 0) | (1 << slot)
        for level, mask in sorted(by_level.items()):
            self._check(_lib.lib().mgcmt_halo_exchange(self.plan._h, level, mask | (_lib.HALO_RING if ring else 0) | (k << 16), None))

    def allreduce_sum(self, values):
        a = np.ascontiguousarray(values, dtype=np.float64).reshape(-1).copy()
        self._check(_lib.lib().mgcmt_allreduce_sum(self.plan._h, _lib.as_dp(a), a.size, None))
        return a

    # -- data ---------------------------------------------------------------------------------------
    def set_shift(self, mu):
        self.set_shifts([float(mu)])

    def set_shifts(self, shifts):
        """one shift per column (vcycle_matrix's shifts=, MGCMTSolver.py:385-388)"""
        self.plan.set_shifts(shifts)
        self.coarse.set_shifts(shifts)

    def upload_local(self, slot, host_rows, vec=0):
        """This rank's rows of a fine-level vector (host array of local_rows*g doubles)."""
        self.plan.upload(0, slot, vec, host_rows)
        self.invalidate(slot)

    def fill_local(self, slot, value, vec=0):
        self.plan.fill(0, slot, vec, value)
        self.invalidate(slot)

    def invalidate(self, slot):
        """Call after anything but vcycle() changed the fine level's V or F: their halo rows travel again."""
        if slot == SLOT_F:
            self._f_halo_valid = False
        if slot == SLOT_V:
            self._v_halo_valid = False

    def download_local(self, slot, vec=0):
        return self.plan.download(0, slot, vec)

    def sync(self):
        self.plan.sync()

    # -- the cycle ----------------------------------------------------------------------------------
    def vcycle(self, nu1, nu2, kind, omega=1.0, nu_coarse=None, k=1, gram_schmidt=False):
        """One V(nu1,nu2) cycle on columns 0..k-1 of V, F of the fine level (sharded down to the switch grid);
        gram_schmidt: vcycle_matrix's modified Gram-Schmidt of the columns on every level on the way up
        (MGCMTSolver.py:434), its inner products all-reduced over the ranks."""
        if kind not in (WJACOBI, GS_MC):
            raise ValueError("only weighted Jacobi and multicolour Gauss-Seidel shard; lexicographic sweeps are sequential")
        if nu1 < 1 or nu2 < 1:
            raise ValueError("the sharded cycle needs at least one pre- and one post-smoothing sweep")
        nu_coarse = nu1 if nu_coarse is None else nu_coarse
        flags = (_lib.SHARDED_V_HALO_VALID if self._v_halo_valid else 0) | (_lib.SHARDED_F_HALO_VALID if self._f_halo_valid else 0)
        if gram_schmidt:
            flags |= _lib.SHARDED_GRAM_SCHMIDT
        self._check(_lib.lib().mgcmt_sharded_vcycle(self.plan._h, self.coarse._h, nu1, nu2, nu_coarse, kind,
                                                    ctypes.c_double(omega), int(k), flags, None))
        self._v_halo_valid = self._f_halo_valid = True

    def residual_norm(self, vec=0):
        """|| F - (A - mu I) V ||_2 of column `vec` over all ranks."""
        P = self.plan
        if not self._v_halo_valid:
            self.exchange_halo((0, SLOT_V), ring=self.emulate is not None, k=self.nvec)
            self._v_halo_valid = True
        P.apply(0, (SLOT_V, vec), (SLOT_T, vec), with_shift=True)
        P.axpy(0, -1.0, (SLOT_F, vec), (SLOT_T, vec))
        local = P.dot(0, (SLOT_T, vec), (SLOT_T, vec))
        return float(self.allreduce_sum([local])[0]) ** 0.5

    def checksum(self, vec=0):
        """(sum V, sum V*V) of column `vec` over all ranks: what a sharded run and the single plan must agree on."""
        P = self.plan
        P.fill(0, SLOT_T, vec, 1.0)
        local = [P.dot(0, (SLOT_V, vec), (SLOT_T, vec)), P.dot(0, (SLOT_V, vec), (SLOT_V, vec))]
        return self.allreduce_sum(local)

    def close(self):
        self.plan.close()
        self.coarse.close()
