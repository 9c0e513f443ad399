"""Host enqueue time vs wall time of the emulated rank share (one GPU): is the sharded cycle host-bound?
usage: rank_share_probe.py [grid] [rank] [of] [switch_grid]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from multigridcmt_amd import _lib, dist_bench
from multigridcmt_amd.distributed import ShardedPlan, rccl_unique_id
from multigridcmt_amd.operators import laplacian_operator

g = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
erank = int(sys.argv[2]) if len(sys.argv) > 2 else 3
of = int(sys.argv[3]) if len(sys.argv) > 3 else 8
sw = int(sys.argv[4]) if len(sys.argv) > 4 else None
op = laplacian_operator(g, "2d") * (-1.0 / np.pi ** 2)
sp = ShardedPlan(op, 8, 0, 1, device=0, switch_grid=sw, transport="rccl", unique_id=rccl_unique_id(), emulate=(erank, of))
sp.set_shift(0.0)
dist_bench.load_rhs(sp, g)
sp.fill_local(_lib.SLOT_V, 0.0)
for opts in ({}, {"overlap": 0}, {"split": 0}):
    sp.set_comm_option(_lib.COMM_OPT_OVERLAP, opts.get("overlap", 1))
    sp.set_comm_option(_lib.COMM_OPT_SPLIT, opts.get("split", 1))
    for _ in range(3):
        sp.vcycle(2, 2, _lib.GS_MC, omega=1.0, nu_coarse=2)
    sp.sync()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        sp.vcycle(2, 2, _lib.GS_MC, omega=1.0, nu_coarse=2)
    t1 = time.perf_counter()
    sp.sync()
    t2 = time.perf_counter()
    print("switch %s options %s: host enqueue %.3f ms/cycle, wall %.3f ms/cycle" % (sp.switch, opts, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3), flush=True)
sp.close()
