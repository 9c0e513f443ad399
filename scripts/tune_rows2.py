"""Fine-level passes of a 16384^2 cycle against the rows a wave marches over (more, shorter chunks = more rounds)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multigridcmt_amd import _lib
from multigridcmt_amd import plan as planmod
from multigridcmt_amd.operators import laplacian_operator
g = 16384
p = planmod.Plan(laplacian_operator(g, "2d") * (-1 / np.pi ** 2), 8, nvec=1)
p.set_shifts([0.0]); p.fill(0, _lib.SLOT_F, 0, 1.0); p.fill(0, _lib.SLOT_V, 0, 0.0)
p.set_option(_lib.OPT_GRAPH, 0)
for rows in (0, 1024, 512, 328, 256, 164, 128, 96, 64):
    p.set_option(_lib.OPT_FUSED_ROWS, rows)
    row = {"rows": rows}
    for name, kind, om in (("wj", _lib.WJACOBI, 2 / 3), ("rb", _lib.GS_MC, 1.0)):
        p.time_smoother(0, kind, 2, om, 3)
        row["%s_pass2_ms" % name] = round(p.time_smoother(0, kind, 2, om, 10) / 10, 4)
        for _ in range(2): p.vcycle(2, 2, kind, omega=om, nu_coarse=2)
        p.sync(); t0 = time.perf_counter()
        for _ in range(8): p.vcycle(2, 2, kind, omega=om, nu_coarse=2)
        p.sync(); row["%s_cycle_ms" % name] = round((time.perf_counter() - t0) / 8 * 1e3, 4)
    print(json.dumps(row), flush=True)
p.set_option(_lib.OPT_FUSED_ROWS, 0)
p.close()
