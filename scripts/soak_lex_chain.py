"""Soak of the chained lexicographic sweeps: random grids, levels, sweep counts, vector counts and smoothers; the chained
launch must give the bits of one launch per sweep, every time.  SOAK_SECONDS (default 150)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multigridcmt_amd import _lib
from multigridcmt_amd.operators import laplacian_operator
from multigridcmt_amd.plan import Plan

budget = float(os.environ.get("SOAK_SECONDS", "150"))
rng = np.random.RandomState(int(os.environ.get("SOAK_SEED", "1")))
t_end = time.time() + budget
plans = {}
cases = mismatches = 0
while time.time() < t_end:
    g = int(2 ** rng.randint(6, 13))            # 64 .. 4096
    k = int(rng.randint(1, 4))
    key = (g, k)
    if key not in plans:
        if len(plans) >= 4:
            plans.pop(next(iter(plans))).close()
        plans[key] = Plan(laplacian_operator(g, "2d") * (-1 / np.pi ** 2), 8, nvec=k)
    p = plans[key]
    level = int(rng.randint(0, max(1, p.num_levels - 2)))
    gl = g >> level
    if gl < 16:
        continue
    nu = int(rng.randint(2, 6))
    kind, omega = (_lib.GS_LEX, 1.0) if rng.rand() < 0.6 else (_lib.SOR_LEX, float(1.0 + rng.rand() * 0.8))
    shifts = rng.rand(k) * 0.5
    p.set_shifts(shifts)
    v0, f = rng.rand(k, gl * gl), rng.rand(k, gl * gl)
    outs = []
    for chain in (1, 0):
        p.set_option(_lib.OPT_LEX_CHAIN, chain)
        for q in range(k):
            p.upload(level, _lib.SLOT_V, q, v0[q]); p.upload(level, _lib.SLOT_F, q, f[q])
        p.smooth(level, kind, nu, omega, k=k)
        outs.append(np.stack([np.array(p.download(level, _lib.SLOT_V, q)) for q in range(k)]))
    cases += 1
    if not np.array_equal(outs[0], outs[1]):
        mismatches += 1
        d = np.abs(outs[0] - outs[1])
        print("MISMATCH g=%d level=%d nu=%d k=%d kind=%d omega=%.3f: %d points, max %.3e" % (g, level, nu, k, kind, omega, int((d > 0).sum()), d.max()), flush=True)
    if cases % 50 == 0:
        print("cases", cases, "mismatches", mismatches, flush=True)
for p in plans.values():
    p.close()
print("SOAK_DONE cases %d mismatches %d" % (cases, mismatches))
sys.exit(1 if mismatches else 0)
