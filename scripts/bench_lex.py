"""Lexicographic Gauss-Seidel / SOR cycles and single sweeps: the wave pipeline (kernels_lexwave.hip) against the
one-workgroup kernel (MGCMT_OPT_LEX_WAVE = 0); "band" = the row-band wavefront (kernels_lexband.hip, option value 2),
"wave" = the skewed column blocks with a scan per row (option value 1).  Prints one JSON line per grid."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multigridcmt_amd import _lib
if os.environ.get("MGCMT_LIB"):
    _lib.use_library(os.environ["MGCMT_LIB"])
from multigridcmt_amd.operators import laplacian_operator
from multigridcmt_amd.plan import Plan


def timed(p, fn, n):
    fn(); fn(); fn(); p.sync(); t0 = time.perf_counter()   # (a cycle's second call instantiates its HIP graph: kept out of the timing)
    for _ in range(n): fn()
    p.sync(); return round((time.perf_counter() - t0) / n * 1e3, 3)


for g in [int(x) for x in os.environ.get("LEX_GRIDS", "512,1024,4096,16384").split(",")]:
    row = {"grid": g}
    for wave in [int(x) for x in os.environ.get("LEX_MODES", "2,1,0").split(",")]:
        if wave == 0 and g > 4096:
            continue
        p = Plan(laplacian_operator(g, "2d") * (-1 / np.pi ** 2), 8, nvec=1)
        p.set_option(_lib.OPT_LEX_WAVE, wave)
        p.set_shifts([0.0]); p.fill(0, _lib.SLOT_F, 0, 1.0); p.fill(0, _lib.SLOT_V, 0, 0.0)
        tag = {2: "band", 1: "wave", 0: "one_wg"}[wave]
        n = 5 if wave else 2
        row["gs_sweep_ms_" + tag] = timed(p, lambda: p.smooth(0, _lib.GS_LEX, 1, 1.0), n)
        if g >= 1024:   # the 9-point Galerkin level below the top (half the rows)
            p.fill(1, _lib.SLOT_F, 0, 1.0); p.fill(1, _lib.SLOT_V, 0, 0.0)
            row["gs_sweep9_ms_" + tag] = timed(p, lambda: p.smooth(1, _lib.GS_LEX, 1, 1.0), n)
        row["gs_V22_ms_" + tag] = timed(p, lambda: p.vcycle(2, 2, _lib.GS_LEX, omega=1.0, nu_coarse=2), n)
        if wave == 1:   # one launch per sweep instead of the sweeps of a smoothing step chained in one launch
            p.set_option(_lib.OPT_LEX_CHAIN, 0)
            row["gs_V22_ms_wave_unchained"] = timed(p, lambda: p.vcycle(2, 2, _lib.GS_LEX, omega=1.0, nu_coarse=2), n)
            p.set_option(_lib.OPT_LEX_CHAIN, 1)
        row["gs_V44_ms_" + tag] = timed(p, lambda: p.vcycle(2, 2, _lib.GS_LEX, omega=1.0, nu_coarse=4), n)
        row["sor_V22_ms_" + tag] = timed(p, lambda: p.vcycle(2, 2, _lib.SOR_LEX, omega=1.3, nu_coarse=2), n)
        p.download(3, _lib.SLOT_V, 0)          # a synchronising call: raises if a block of the pipeline gave up
        p.close()
    if "gs_V22_ms_one_wg" in row:
        if "gs_V22_ms_wave" in row:
            row["speedup_V22"] = round(row["gs_V22_ms_one_wg"] / row["gs_V22_ms_wave"], 1)
        if "gs_V22_ms_band" in row:
            row["speedup_V22_band"] = round(row["gs_V22_ms_one_wg"] / row["gs_V22_ms_band"], 1)
    print(json.dumps(row), flush=True)
