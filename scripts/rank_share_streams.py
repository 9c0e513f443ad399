"""Does the staged-transfer machinery (8 DMA streams + pinned ring of csrc/transfer.hip, created by the first large
upload) change the emulated rank share?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from multigridcmt_amd import _lib, dist_bench
from multigridcmt_amd.operators import laplacian_operator
from multigridcmt_amd.plan import Plan


def share(tag):
    print(tag, "%.3f" % dist_bench.time_rank_share(32768, 2, 8, "rb", 3, 8)["ms_per_rank_share"], flush=True)


share("fresh (its own right-hand side upload is 268 MB: staged)")
g = 8192
plan = Plan(laplacian_operator(g, "2d") * (-1.0 / np.pi ** 2), 8, nvec=1, device=0)
plan.set_shifts([0.0])
plan.upload(0, _lib.SLOT_F, 0, np.ones(g * g))
share("after a 512 MB upload")
x = plan.download(0, _lib.SLOT_F, 0)
share("after a 512 MB download (pinned result buffer)")
d = plan.dot(0, (_lib.SLOT_F, 0), (_lib.SLOT_F, 0))
share("after a dot")
plan.close()
share("after close")
