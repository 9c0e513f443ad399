"""Chained against unchained lexicographic sweeps at size, level by level: where, and by how much, do the results differ?
MGCMT_LIB selects the library (variants/lib_*.so)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multigridcmt_amd import _lib
if os.environ.get("MGCMT_LIB"):
    _lib.use_library(os.environ["MGCMT_LIB"])
from multigridcmt_amd.operators import laplacian_operator
from multigridcmt_amd.plan import Plan
g = int(os.environ.get("LEX_GRID", "16384"))
p = Plan(laplacian_operator(g, "2d") * (-1 / np.pi ** 2), 8, nvec=1)
p.set_shifts([0.0])
rng = np.random.RandomState(5)
for level in range(0, p.num_levels - 1):
    gl = g >> level
    if gl < 16:
        break
    f, v0 = rng.rand(gl * gl), rng.rand(gl * gl)
    p.upload(level, _lib.SLOT_F, 0, f)
    for nu in (2, 4):
        for kind, omega in ((_lib.GS_LEX, 1.0), (_lib.SOR_LEX, 1.3)):
            outs = []
            for chain in (0, 1, 1):
                p.set_option(_lib.OPT_LEX_CHAIN, chain)
                p.upload(level, _lib.SLOT_V, 0, v0)
                p.smooth(level, kind, nu, omega)
                outs.append(np.array(p.download(level, _lib.SLOT_V, 0)).reshape(gl, gl))
            for t in (1, 2):
                d = np.abs(outs[t] - outs[0])
                bad = np.argwhere(d > 0)
                tag = "level %d (%d^2) nu %d kind %d trial %d:" % (level, gl, nu, kind, t)
                if len(bad) == 0:
                    print(tag, "identical", flush=True)
                else:
                    rows, cols = np.unique(bad[:, 0]), np.unique(bad[:, 1])
                    print(tag, "DIFFER at %d points, max %.3e; rows %d..%d (%d) cols %d..%d (%d); first %s" % (
                        len(bad), d.max(), rows[0], rows[-1], len(rows), cols[0], cols[-1], len(cols), bad[:4].tolist()), flush=True)
p.close()
