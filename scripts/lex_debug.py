"""Diagnostic run of the lexicographic wave pipeline (library built with -DMGCMT_LEXWAVE_DEBUG, variants/lib_lexdebug.so):
per block the time spent, rows and slow-path entries of ONE fine-level Gauss-Seidel sweep."""
import ctypes, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multigridcmt_amd import _lib
_lib.use_library(os.path.join(ROOT, "variants", "lib_lexdebug.so"))
from multigridcmt_amd.operators import laplacian_operator
from multigridcmt_amd.plan import Plan
for g in (1024, 4096):
    p = Plan(laplacian_operator(g, "2d") * (-1 / np.pi ** 2), g, nvec=1)
    p.set_shifts([0.0]); p.fill(0, _lib.SLOT_F, 0, 1.0); p.fill(0, _lib.SLOT_V, 0, 0.0)
    for _ in range(2):
        p.smooth(0, _lib.GS_LEX, 1, 1.0)
    nb = (2 * g - 1 + 63) // 64
    out = (ctypes.c_uint32 * (2 + 4 * nb))()
    _lib.check(_lib.lib().mgcmt_lex_wave_stats(p._h, out, len(out)))
    a = np.array(out[2:], dtype=np.int64).reshape(nb, 4)
    t0 = a[:, 3].min()
    rows = []
    for J in (0, 1, 2, 3, nb // 4, nb // 2, 3 * nb // 4, nb - 2, nb - 1):
        ticks, nrows, slow, start = a[J]
        rows.append({"J": int(J), "rows": int(nrows), "us": ticks / 100.0, "us_per_row": round(ticks / 100.0 / max(nrows, 1), 3), "slow": int(slow),
                     "start_us": ((start - t0) & 0xffffffff) / 100.0})
    print(json.dumps({"grid": g, "blocks": nb, "started": int(out[0]), "error": int(out[1]),
                      "mean_us_per_row": round(float((a[:, 0] / 100.0).sum() / a[:, 1].sum()), 3),
                      "slow_total": int(a[:, 2].sum()), "rows_total": int(a[:, 1].sum()), "blocks_detail": rows}), flush=True)
    p.close()
