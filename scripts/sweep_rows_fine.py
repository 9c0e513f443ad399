"""Rows per chunk on the 16384^2 fine level: the cycle's down pass (mode 2|8), up pass (mode 1|32) and the plain pass, HIP
events (mgcmt_time_fused_pass).  Short chunks = many short-lived waves dispatched in address order (the regime in which a
plain copy reaches 6.3 TB/s, profiles/r03_probe_copy.txt) against one round of long-lived waves."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multigridcmt_amd import _lib
from multigridcmt_amd.operators import laplacian_operator
from multigridcmt_amd.plan import Plan
if os.environ.get("MGCMT_LIBRARY"):                      # a compile-time variant of the library (scripts/build_variants.sh)
    _lib.use_library(os.environ["MGCMT_LIBRARY"])
g = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
p = Plan(laplacian_operator(g, "2d") * (-1 / np.pi ** 2), 8, nvec=1)
p.set_shifts([0.0])
p.fill(0, _lib.SLOT_F, 0, 1.0); p.fill(0, _lib.SLOT_V, 0, 0.0)
n = float(g) * g
for kind, name, om in ((_lib.WJACOBI, "wjacobi", 2 / 3), (_lib.GS_MC, "rb", 1.0)):
    for rows in [int(x) for x in os.environ.get("SWEEP_ROWS", "0,32,64,128,256,1024").split(",")]:
        p.set_option(_lib.OPT_FUSED_ROWS, rows)
        rec = {"smoother": name, "rows": rows}
        for label, mode, bpp in (("down", 2 | 8, 18), ("up", 1 | 32, 26), ("plain", 0, 24)):
            ms = p.time_fused_pass(0, kind, 2, om, mode, 10)
            rec[label + "_ms"] = round(ms, 4); rec[label + "_frac"] = round(n * bpp / ms / 1e6 / 8000, 3)
        print(json.dumps(rec), flush=True)
p.close()
