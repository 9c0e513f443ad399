"""Child process of tests/test_full_size_gpu.py::test_sharded_plan_single_rank_on_gpu (GPU box only): the sharded
cycle of libmgcmt_hip.so at world size 1 through its RCCL transport.

1. ring self-exchange: mgcmt_halo_exchange with the rank as its own neighbour — RCCL send/recv to self must put the
   bottom rows into the upper halo rows and the top rows into the lower halo rows (read back through torch views).
2. the sharded cycle in self-ring mode: boundary rows first, their RCCL exchange on the second stream beside the
   interior launch, the all-gather replaced by its one-rank copy — the result must be the single plan's.
3. the same through the external transport (torch.distributed, backend nccl = RCCL) at world size 1.
"""
import os
import socket
import sys

import numpy as np
import torch                      # before libmgcmt_hip.so: one HIP runtime per process
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multigridcmt_amd import _lib                                   # noqa: E402
from multigridcmt_amd.distributed import ShardedPlan, _DevicePointer, rccl_unique_id   # noqa: E402
from multigridcmt_amd.operators import laplacian_operator            # noqa: E402
from multigridcmt_amd.plan import Plan                               # noqa: E402

s = socket.socket()
s.bind(("127.0.0.1", 0))
port = s.getsockname()[1]
s.close()
os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
g = 4096
op = laplacian_operator(g, "2d") * (-1 / np.pi ** 2)
f = np.random.RandomState(2).rand(g * g)

p = Plan(op, 8, nvec=1)
p.set_shifts([0.0])
p.upload(0, _lib.SLOT_F, 0, f)
p.fill(0, _lib.SLOT_V, 0, 0.0)
for _ in range(2):
    p.vcycle(2, 2, _lib.GS_MC, omega=1.0, nu_coarse=2)
want = p.download(0, _lib.SLOT_V, 0)
p.apply(0, (_lib.SLOT_V, 0), (_lib.SLOT_T, 0), with_shift=True)
p.axpy(0, -1.0, (_lib.SLOT_F, 0), (_lib.SLOT_T, 0))
want_res = np.sqrt(p.dot(0, (_lib.SLOT_T, 0), (_lib.SLOT_T, 0)))
p.close()

for transport in ("rccl", "torch"):
    sp = ShardedPlan(op, 8, 0, 1, device=0, switch_grid=1024, transport=transport,
                     unique_id=rccl_unique_id() if transport == "rccl" else None)
    sp.set_shift(0.0)
    sp.upload_local(_lib.SLOT_F, f)
    sp.fill_local(_lib.SLOT_V, 0.0)
    if transport == "rccl":
        # 1. ring self-exchange on levels 0 and 1 of the spare slot W (level 1 of a 4096^2 / 1024 plan is a strip level too)
        W = _lib.SLOT_W
        sp.plan.upload(0, W, 0, np.random.RandomState(3).rand(g * g))
        sp.plan.upload(1, W, 0, np.arange((g // 2) ** 2, dtype=np.float64))
        sp.exchange_halo((0, W), ring=True)
        sp.exchange_halo((1, W), ring=True)
        sp.sync()
        for level, cols in ((0, g), (1, g // 2)):
            rows = cols
            halo, H = sp.plan.level_halo(level)         # rows kept / rows exchanged (8 behind the 5-point operator, 10 behind the 9-point one)
            assert (halo, H) == (_lib.HALO_ROWS, 8 if level == 0 else 10), (halo, H)
            base = sp.plan.vec_ptr(level, W, 0) - H * cols * 8
            flat = torch.as_tensor(_DevicePointer(base, (rows + 2 * H) * cols), device="cuda:0").cpu().numpy().reshape(rows + 2 * H, cols)
            assert np.array_equal(flat[:H], flat[rows:rows + H]), "upper halo rows != bottom rows (level %d)" % level
            assert np.array_equal(flat[rows + H:], flat[H:2 * H]), "lower halo rows != top rows (level %d)" % level
        print("ring self-exchange through RCCL ok")
        # 2. the cycle with the rank as its own neighbour: split launches, RCCL on the second stream, overlap — the rows
        #    received land in halo rows the fused kernels of a whole-grid level never read, so the result is unchanged
        sp.set_comm_option(_lib.COMM_OPT_SELF_RING, 1)
        sp.set_comm_option(_lib.COMM_OPT_SPLIT, 2)
        for _ in range(2):
            sp.vcycle(2, 2, _lib.GS_MC, omega=1.0, nu_coarse=2)
        ring_result = sp.download_local(_lib.SLOT_V)
        assert np.array_equal(ring_result, want), "self-ring cycle differs from the single plan"
        for opt in (_lib.COMM_OPT_OVERLAP, _lib.COMM_OPT_SPLIT):          # overlap off / split off: the same numbers
            sp.set_comm_option(opt, 0)
            sp.fill_local(_lib.SLOT_V, 0.0)
            for _ in range(2):
                sp.vcycle(2, 2, _lib.GS_MC, omega=1.0, nu_coarse=2)
            assert np.array_equal(sp.download_local(_lib.SLOT_V), want), opt
        print("self-ring cycles (overlap / no overlap / no split) bit-equal to the single plan")
        # back to the plain one-rank chain; the ring traffic left the neighbours' rows in halo rows that the Dirichlet
        # kernels expect to be zero: clear them
        sp.set_comm_option(_lib.COMM_OPT_SELF_RING, 0)
        sp.set_comm_option(_lib.COMM_OPT_OVERLAP, 1)
        sp.set_comm_option(_lib.COMM_OPT_SPLIT, 1)
        for level in range(sp.strip_levels + 1):
            for slot in (_lib.SLOT_V, _lib.SLOT_F, _lib.SLOT_T):
                sp.plan.zero(level, slot, 0)
        sp.upload_local(_lib.SLOT_F, f)
        sp.fill_local(_lib.SLOT_V, 0.0)
    for _ in range(2):
        sp.vcycle(2, 2, _lib.GS_MC, omega=1.0, nu_coarse=2)
    got, res = sp.download_local(_lib.SLOT_V), sp.residual_norm()
    err = np.linalg.norm(got - want) / np.linalg.norm(want)
    print(transport, "rel err", err, "residual", res, want_res, "strip levels", sp.strip_levels)
    assert err < 1e-12 and abs(res - want_res) < 1e-9 * want_res
    sp.close()
# 4. rank emulation (MGCMT_COMM_OPT_EMULATE_OF): the strip of an INTERIOR rank of an 8-rank job with itself as both neighbours —
#    unlike the whole-grid plans above its passes READ the halo rows the exchanges fill, so a missing dependency between the
#    two lanes (interior launches on the main stream, edge rows + exchange on the exchange stream) would show as a difference
#    between the overlapped schedule and the in-order one.  Bit for bit, three cycles, 9-point strip levels included.
results = {}
for overlap, split in ((1, 2), (0, 2), (0, 0)):
    sp = ShardedPlan(op, 8, 0, 1, device=0, switch_grid=512, transport="rccl", unique_id=rccl_unique_id(), emulate=(3, 8))
    assert sp.strip_levels == 3 and [sp.plan.level_halo(l)[1] for l in range(3)] == [8, 10, 10]
    sp.set_shift(0.3)
    sp.set_comm_option(_lib.COMM_OPT_OVERLAP, overlap)
    sp.set_comm_option(_lib.COMM_OPT_SPLIT, split)
    rows = g // 8
    sp.upload_local(_lib.SLOT_F, f[3 * rows * g:4 * rows * g])
    sp.fill_local(_lib.SLOT_V, 0.0)
    for _ in range(3):
        sp.vcycle(2, 2, _lib.GS_MC, omega=1.0, nu_coarse=2)
    results[(overlap, split)] = sp.download_local(_lib.SLOT_V)
    assert np.all(np.isfinite(results[(overlap, split)]))
    sp.close()
assert np.array_equal(results[(1, 2)], results[(0, 2)]), "two-lane schedule differs from the in-order split schedule"
assert np.array_equal(results[(1, 2)], results[(0, 0)]), "split schedule differs from the unsplit one"
print("emulated interior rank: overlapped / in-order / unsplit schedules bit-equal")
dist.destroy_process_group()
print("SHARDED_WORLD1_OK")
