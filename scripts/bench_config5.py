"""Config 5 timings on the GPU: potential-well operator (three Kronecker terms -> Op9<3> fused kernels on every level)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigridcmt_amd import _lib, drivers
from multigridcmt_amd.operators import potential_well_operator
from multigridcmt_amd.plan import Plan

g = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
out = {"grid": g}
op = potential_well_operator(g, 50.0, (g // 4, 3 * g // 4))
p = Plan(op, 8, nvec=1)
p.set_shifts([0.0])
rng = np.random.RandomState(0)
p.upload(0, _lib.SLOT_F, 0, rng.rand(g * g))
for name, kind, omega in (("wjacobi", _lib.WJACOBI, 2. / 3.), ("rb", _lib.GS_MC, 1.0)):
    for fused in (1, 0):
        p.set_option(_lib.OPT_FUSED, fused)
        p.zero(0, _lib.SLOT_V, 0)
        for _ in range(3):
            p.vcycle(2, 2, kind, omega=omega, nu_coarse=2)
        p.sync()
        t = time.perf_counter()
        for _ in range(10):
            p.vcycle(2, 2, kind, omega=omega, nu_coarse=2)
        p.sync()
        out["vcycle_ms_%s_%s" % (name, "fused" if fused else "baseline")] = (time.perf_counter() - t) * 100
    p.set_option(_lib.OPT_FUSED, 1)
    ms = p.time_smoother(0, kind, 2, omega, 10) / 10
    out["smooth2_ms_%s" % name] = ms
    out["smooth2_GBs_24B_%s" % name] = 2 * 24.0 * g * g / ms / 1e6
p.close()
for method, cycles in (("vcycle", 10), ("rqmg", 3)):
    hist = []
    drivers.potential_well_eigensolve(g, cycles=1, method=method)          # plan creation, graph capture
    t = time.perf_counter()
    stats = {}
    rho, _ = drivers.potential_well_eigensolve(g, cycles=cycles, method=method, history=hist, stats=stats)
    out["%s_seconds_%d_cycles" % (method, cycles)] = time.perf_counter() - t
    if "loop_seconds" in stats:
        out["%s_ms_per_iteration" % method] = stats["loop_seconds"] / cycles * 1e3
    out["%s_rho" % method] = hist
print(json.dumps(out, indent=1))
