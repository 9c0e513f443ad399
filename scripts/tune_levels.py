"""Times fused passes on levels 0..2 of a 16384^2 plan for each library variant (GPU box)."""
import glob, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multigridcmt_amd import _lib
from multigridcmt_amd import plan as planmod
from multigridcmt_amd.operators import laplacian_operator
g = 16384
libs = [_lib.DEFAULT_LIBRARY] + sorted(glob.glob(os.path.join(ROOT, "variants", "lib_*.so")))
for path in libs:
    _lib.use_library(path)
    p = planmod.Plan(laplacian_operator(g, "2d") * (-1 / np.pi ** 2), 8, nvec=1)
    p.set_shifts([0.0])
    for l in range(4):
        p.fill(l, _lib.SLOT_F, 0, 1.0); p.fill(l, _lib.SLOT_V, 0, 0.0)
    row = {"lib": os.path.basename(path)}
    for l in range(4):
        for name, kind, om in (("wj", _lib.WJACOBI, 2 / 3), ("rb", _lib.GS_MC, 1.0)):
            p.time_smoother(l, kind, 2, om, 3)
            ms = p.time_smoother(l, kind, 2, om, 20) / 20
            n = (g >> l) ** 2
            row["L%d_%s2_ms" % (l, name)] = round(ms, 4)
            row["L%d_%s2_GBs" % (l, name)] = round(n * 24 * (1 if (name == "wj" or l == 0) else 2) / (ms * 1e-3) / 1e9)
    # whole cycles
    for name, kind, om in (("wj", _lib.WJACOBI, 2 / 3), ("rb", _lib.GS_MC, 1.0)):
        import time
        for _ in range(2): p.vcycle(2, 2, kind, omega=om, nu_coarse=2)
        p.sync(); t0 = time.perf_counter()
        for _ in range(10): p.vcycle(2, 2, kind, omega=om, nu_coarse=2)
        p.sync(); row["cycle_%s_ms" % name] = round((time.perf_counter() - t0) * 100, 3)
    p.close()
    print(json.dumps(row), flush=True)
