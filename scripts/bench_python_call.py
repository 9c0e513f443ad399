"""The drop-in API with host arrays in and out (numpy -> H2D -> cycle -> D2H): PCIe-inclusive time per call."""
import cProfile, io, json, os, pstats, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multigridcmt_amd import MGCMTSolver, MGCMTStencilMaker
solver, sm = MGCMTSolver(), MGCMTStencilMaker()
for g in (1024, 4096, 8192):
    A = sm.laplacian(g, dimension="2d", matrix_free=True) * (-1 / np.pi ** 2)
    rng = np.random.RandomState(0)
    f, v0 = rng.rand(g * g), np.zeros(g * g)
    t0 = time.perf_counter()
    w = solver.vcycle(v0, f, A, sm, nu1=2, nu2=2, smoother=solver.wjacobi, dimension="2d", lowest_level=8)
    cold = time.perf_counter() - t0            # plan set-up, kernels loaded, first page-locked result buffer
    for _ in range(2):                         # (the second result buffer: `w` holds one while the next call fills another)
        w = solver.vcycle(v0, f, A, sm, nu1=2, nu2=2, smoother=solver.wjacobi, dimension="2d", lowest_level=8)
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        w = solver.vcycle(v0, f, A, sm, nu1=2, nu2=2, smoother=solver.wjacobi, dimension="2d", lowest_level=8)
    dt = (time.perf_counter() - t0) / n
    row = {"g": g, "first_call_ms": round(cold * 1e3, 1), "python_vcycle_call_ms": round(dt * 1e3, 2), "bytes_moved_MB": round(3 * 8 * g * g / 1e6, 1),
           "effective_GBs": round(3 * 8 * g * g / dt / 1e9, 2)}
    # the iteration the reference's drivers run: the result is the next call's start value (1DPotMatrixVcycle.py:60-70)
    t0 = time.perf_counter()
    for _ in range(n):
        w = solver.vcycle(w, f, A, sm, nu1=2, nu2=2, smoother=solver.wjacobi, dimension="2d", lowest_level=8)
    dt = (time.perf_counter() - t0) / n
    row["iterated_call_ms"] = round(dt * 1e3, 2)
    row["iterated_GBs"] = round(3 * 8 * g * g / dt / 1e9, 2)
    print(json.dumps(row), flush=True)
g = 4096
A = sm.laplacian(g, dimension="2d", matrix_free=True) * (-1 / np.pi ** 2)
f, v0 = np.random.RandomState(0).rand(g * g), np.zeros(g * g)
pr = cProfile.Profile(); pr.enable()
for _ in range(3):
    solver.vcycle(v0, f, A, sm, nu1=2, nu2=2, smoother=solver.wjacobi, dimension="2d", lowest_level=8)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(14); print(s.getvalue()[:3500])
