"""Operator application (row march) and the Ritz pass at size, for every library under variants/ (GPU box)."""
import glob, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multigridcmt_amd import _lib
from multigridcmt_amd import plan as planmod
from multigridcmt_amd.operators import laplacian_operator, potential_well_operator
libs = [_lib.DEFAULT_LIBRARY] + sorted(glob.glob(os.path.join(ROOT, "variants", "lib_*.so")))
for path in libs:
    _lib.use_library(path)
    row = {"lib": os.path.basename(path)}
    for name, g, op in (("well_8192", 8192, potential_well_operator(8192, 50.0, (2048, 6144))), ("laplacian_16384", 16384, laplacian_operator(16384, "2d") * (-1 / np.pi ** 2))):
        p = planmod.Plan(op, 8, nvec=2)
        p.set_shifts([0.0, 0.3])
        X, W, S = (_lib.SLOT_W, 1), (_lib.SLOT_V, 0), (_lib.SLOT_W, 0)
        p.fill(0, X[0], X[1], 1.0); p.fill(0, W[0], W[1], 0.5)
        for what, fn in (("apply", lambda: p.apply(0, X, S, with_shift=True)), ("ritz", lambda: p.ritz_pair(0, X, W, S))):
            for _ in range(3): fn()
            p.sync(); t0 = time.perf_counter()
            for _ in range(20): fn()
            p.sync(); ms = (time.perf_counter() - t0) / 20 * 1e3
            row["%s_%s_ms" % (what, name)] = round(ms, 4)
            row["%s_%s_TBs" % (what, name)] = round(16.0 * g * g / ms / 1e9, 2)
        p.close()
    print(json.dumps(row), flush=True)
