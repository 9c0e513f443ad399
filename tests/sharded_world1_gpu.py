"""Child process of tests/test_full_size_gpu.py::test_sharded_plan_single_rank_on_gpu (GPU box only)."""
import os
import socket
import sys

import numpy as np
import torch                      # before libmgcmt_hip.so: one HIP runtime per process
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multigridcmt_amd import _lib                                   # noqa: E402
from multigridcmt_amd.distributed import ShardedPlan                 # noqa: E402
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
sp = ShardedPlan(op, 8, 0, 1, device=0, switch_grid=1024)
sp.set_shift(0.0)
sp.upload_local(_lib.SLOT_F, f)
sp.plan.fill(0, _lib.SLOT_V, 0, 0.0)
for _ in range(2):
    sp.vcycle(2, 2, _lib.GS_MC, omega=1.0, nu_coarse=2)
got, res = sp.download_local(_lib.SLOT_V), sp.residual_norm()
sp.close()
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
err = np.linalg.norm(got - want) / np.linalg.norm(want)
print("rel err", err, "residual", res, want_res, "strip levels", sp.strip_levels)
assert err < 1e-12 and abs(res - want_res) < 1e-9 * want_res
dist.destroy_process_group()
print("SHARDED_WORLD1_OK")
