"""Times the fine-level fused smoother pass for library variants / rows-per-chunk settings (GPU box)."""
import glob, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multigridcmt_amd import _lib
from multigridcmt_amd import plan as planmod
from multigridcmt_amd.operators import laplacian_operator

g = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
libs = [_lib.DEFAULT_LIBRARY] + sorted(glob.glob(os.path.join(ROOT, "build", "variants", "lib_*.so")))
f = np.random.RandomState(1).rand(g * g)
for path in libs:
    _lib.use_library(path)
    p = planmod.Plan(laplacian_operator(g, "2d") * (-1 / np.pi ** 2), g // 2, nvec=1)
    p.set_shifts([0.0]); p.upload(0, _lib.SLOT_F, 0, f); p.fill(0, _lib.SLOT_V, 0, 0.0)
    for rows in (0, 64, 128, 256, 512, 1024, 2048):
        p.set_option(_lib.OPT_FUSED_ROWS, rows)
        row = {"lib": os.path.basename(path), "rows": rows}
        for name, kind, om in (("wj", _lib.WJACOBI, 2 / 3), ("rb", _lib.GS_MC, 1.0)):
            for nu in (1, 2):
                p.time_smoother(0, kind, nu, om, 3)
                ms = p.time_smoother(0, kind, nu, om, 20) / 20
                row["%s%d_ms" % (name, nu)] = round(ms, 4)
        row["wj2_TBs"] = round(g * g * 24 / (row["wj2_ms"] * 1e-3) / 1e12, 3)
        print(json.dumps(row), flush=True)
    p.set_option(_lib.OPT_FUSED_ROWS, 0)
    p.close()
