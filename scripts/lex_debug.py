"""Diagnostic run of the lexicographic wave pipeline (library built with -DMGCMT_LEXWAVE_DEBUG, variants/lib_lexdebug.so):
per block the time spent, rows and slow-path entries of a Gauss-Seidel smoothing step.  usage: lex_debug.py [level] [sweeps]
(level 1 = the 9-point Galerkin level; sweeps > 1 = chained: the words hold whichever sweep's block wrote last)"""
import ctypes, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multigridcmt_amd import _lib
_lib.use_library(os.path.join(ROOT, "variants", "lib_lexdebug.so"))
from multigridcmt_amd.operators import laplacian_operator
from multigridcmt_amd.plan import Plan
LEVEL = int(sys.argv[1]) if len(sys.argv) > 1 else 0
SWEEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for g in (1024, 4096):
    p = Plan(laplacian_operator(g, "2d") * (-1 / np.pi ** 2), g >> LEVEL, nvec=1)
    p.set_shifts([0.0]); p.fill(LEVEL, _lib.SLOT_F, 0, 1.0); p.fill(LEVEL, _lib.SLOT_V, 0, 0.0)
    import time
    for _ in range(3):
        p.sync(); t0 = time.perf_counter()
        p.smooth(LEVEL, _lib.GS_LEX, SWEEPS, 1.0)
        p.sync(); ms = (time.perf_counter() - t0) * 1e3
    gl = g >> LEVEL
    nb = (2 * gl - 1 + 63) // 64
    out = (ctypes.c_uint32 * (2 + 4 * nb))()
    _lib.check(_lib.lib().mgcmt_lex_wave_stats(p._h, out, len(out)))
    a = np.array(out[2:], dtype=np.int64).reshape(nb, 4)
    t0 = a[:, 3].min()
    rows = []
    for J in (0, 1, 2, 3, nb // 4, nb // 2, 3 * nb // 4, nb - 2, nb - 1):
        ticks, nrows, slow, start = a[J]
        rows.append({"J": int(J), "rows": int(nrows), "us": ticks / 100.0, "us_per_row": round(ticks / 100.0 / max(nrows, 1), 3), "slow": int(slow),
                     "start_us": ((start - t0) & 0xffffffff) / 100.0})
    print(json.dumps({"grid": gl, "level": LEVEL, "sweeps": SWEEPS, "ms_with_sync": round(ms, 3), "blocks": nb, "started": int(out[0]), "error": int(out[1]),
                      "mean_us_per_row": round(float((a[:, 0] / 100.0).sum() / a[:, 1].sum()), 3),
                      "slow_total": int(a[:, 2].sum()), "rows_total": int(a[:, 1].sum()), "blocks_detail": rows}), flush=True)
    p.close()
