"""Secondary timings for DESIGN.md: lexicographic (parity-mode) Gauss-Seidel cycles, 1-D cycles, rqmin."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multigridcmt_amd import _lib
from multigridcmt_amd.operators import laplacian_operator
from multigridcmt_amd.plan import Plan
def cyc(p, kind, om, nu, n):
    for _ in range(2): p.vcycle(nu, nu, kind, omega=om, nu_coarse=nu)
    p.sync(); t0 = time.perf_counter()
    for _ in range(n): p.vcycle(nu, nu, kind, omega=om, nu_coarse=nu)
    p.sync(); return round((time.perf_counter() - t0) / n * 1e3, 3)
for g in (256, 1024, 4096):
    p = Plan(laplacian_operator(g, "2d") * (-1 / np.pi ** 2), 8, nvec=1)
    p.set_shifts([0.0]); p.fill(0, _lib.SLOT_F, 0, 1.0); p.fill(0, _lib.SLOT_V, 0, 0.0)
    print(json.dumps({"2d": g, "gs_lex_V22_ms": cyc(p, _lib.GS_LEX, 1.0, 2, 3), "sor_lex_1.3_V22_ms": cyc(p, _lib.SOR_LEX, 1.3, 2, 3)}), flush=True)
    p.close()
for n in (1024, 1 << 16, 1 << 20, 1 << 24):
    p = Plan(laplacian_operator(n, "1d") * (-1 / np.pi ** 2), 8, nvec=1)
    p.set_shifts([0.0]); p.fill(0, _lib.SLOT_F, 0, 1.0); p.fill(0, _lib.SLOT_V, 0, 0.0)
    print(json.dumps({"1d": n, "wj_V22_ms": cyc(p, _lib.WJACOBI, 2 / 3, 2, 5), "rb_V22_ms": cyc(p, _lib.GS_MC, 1.0, 2, 5),
                      "gs_lex_V22_ms": cyc(p, _lib.GS_LEX, 1.0, 2, 3)}), flush=True)
    p.close()
