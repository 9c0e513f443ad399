import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multigridcmt_amd import _lib
from multigridcmt_amd import plan as planmod
from multigridcmt_amd.operators import laplacian_operator
for g in (16384, 4096):
    p = planmod.Plan(laplacian_operator(g, "2d") * (-1 / np.pi ** 2), g // 2, nvec=1)
    p.set_shifts([0.0]); p.fill(0, _lib.SLOT_F, 0, 1.0); p.fill(0, _lib.SLOT_V, 0, 0.0)
    for rows in (0, 32, 64, 128, 256, 512, 600, 700, 746, 800, 1024):
        p.set_option(_lib.OPT_FUSED_ROWS, rows)
        row = {"g": g, "rows": rows}
        for name, kind, om in (("wj", _lib.WJACOBI, 2 / 3), ("rb", _lib.GS_MC, 1.0)):
            for nu in (1, 2):
                p.time_smoother(0, kind, nu, om, 3)
                row["%s%d_ms" % (name, nu)] = round(p.time_smoother(0, kind, nu, om, 20) / 20, 4)
        print(json.dumps(row), flush=True)
    p.set_option(_lib.OPT_FUSED_ROWS, 0)
    p.close()
