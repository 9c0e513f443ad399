"""vcycle_matrix (k columns, Gram-Schmidt on the way up) timings: device-only cycles and the full Python call."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multigridcmt_amd import MGCMTSolver, MGCMTStencilMaker, _lib
from multigridcmt_amd.operators import laplacian_operator
from multigridcmt_amd.plan import Plan
k = 10
for g in (64, 256, 1024, 4096):
    op = laplacian_operator(g, "2d") * (-1 / np.pi ** 2)
    p = Plan(op, 8, nvec=k)
    p.set_shifts(np.linspace(1.9, 9.8, k))
    rng = np.random.RandomState(0)
    for q in range(k):
        p.upload(0, _lib.SLOT_F, q, rng.rand(g * g)); p.fill(0, _lib.SLOT_V, q, 0.0)
    for _ in range(3): p.vcycle(4, 4, _lib.WJACOBI, omega=2 / 3, k=k, nu_coarse=4, gram_schmidt=True)
    p.sync(); t0 = time.perf_counter(); n = 20 if g <= 1024 else 5
    for _ in range(n): p.vcycle(4, 4, _lib.WJACOBI, omega=2 / 3, k=k, nu_coarse=4, gram_schmidt=True)
    p.sync(); dev = (time.perf_counter() - t0) / n
    p.close()
    row = {"g": g, "k": k, "device_cycle_ms": round(dev * 1e3, 3)}
    if g <= 1024:
        solver, sm = MGCMTSolver(), MGCMTStencilMaker()
        A = sm.laplacian(g, dimension="2d", matrix_free=True) * (-1 / np.pi ** 2)
        F = rng.rand(g * g, k); W0 = np.zeros((g * g, k)); sh = np.linspace(1.9, 9.8, k)
        solver.vcycle_matrix(W0, F, A, sm, shifts=sh, dimension="2d", lowest_level=8)
        t0 = time.perf_counter()
        for _ in range(5): solver.vcycle_matrix(W0, F, A, sm, shifts=sh, dimension="2d", lowest_level=8)
        row["python_call_ms"] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
    print(json.dumps(row), flush=True)
