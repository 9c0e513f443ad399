"""Child process of tests/test_full_size_gpu.py::test_sharded_ranks_share_one_gpu (GPU box only): one rank of a
gloo group whose ranks all compute on the box's single GPU, with the halo rows staged through host memory.
usage: sharded_gloo_gpu.py RANK WORLD PORT GRID KIND OUTDIR"""
import os
import sys

import numpy as np
import torch                      # before libmgcmt_hip.so: one HIP runtime per process
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multigridcmt_amd import _lib                                   # noqa: E402
from multigridcmt_amd.distributed import ShardedPlan                 # noqa: E402
from multigridcmt_amd.operators import laplacian_operator            # noqa: E402

rank, world, port, g, kind_name, out_dir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), sys.argv[5], sys.argv[6]
os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", port
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
kind, omega = (_lib.WJACOBI, 2. / 3.) if kind_name == "wjacobi" else (_lib.GS_MC, 1.0)
op = laplacian_operator(g, "2d") * (-1 / np.pi ** 2)
sp = ShardedPlan(op, 8, rank, world, device=0, switch_grid=g // 4)
assert sp._torch_transport.stage_host
sp.plan.set_option(_lib.OPT_RECOMPUTE, 2)
sp.set_shift(0.3)
sp.set_comm_option(_lib.COMM_OPT_SPLIT, 2)
rng = np.random.RandomState(9)
f, v0 = rng.rand(g * g), rng.rand(g * g)
rows = g // world
sl = slice(rank * rows * g, (rank + 1) * rows * g)
sp.upload_local(_lib.SLOT_F, f[sl])
sp.upload_local(_lib.SLOT_V, v0[sl])
for _ in range(2):
    sp.vcycle(2, 2, kind, omega=omega, nu_coarse=2)
res = sp.residual_norm()
np.save(os.path.join(out_dir, "part%d.npy" % rank), sp.download_local(_lib.SLOT_V))
if rank == 0:
    np.save(os.path.join(out_dir, "res.npy"), np.array([res, sp.strip_levels]))
sp.close()
dist.destroy_process_group()
print("SHARDED_GLOO_GPU_OK", rank)
