import json, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from multigridcmt_amd import _lib
from multigridcmt_amd.operators import laplacian_operator
from multigridcmt_amd.plan import Plan
k = 10
for g in (1024, 4096):
    op = laplacian_operator(g, "2d") * (-1 / np.pi ** 2)
    p = Plan(op, 8, nvec=k)
    p.set_shifts(np.linspace(1.9, 9.8, k))
    rng = np.random.RandomState(0)
    for q in range(k):
        p.upload(0, _lib.SLOT_F, q, rng.rand(g * g)); p.fill(0, _lib.SLOT_V, q, 0.0)
    row = {"g": g}
    for gs in (True, False):
        for _ in range(3): p.vcycle(4, 4, _lib.WJACOBI, omega=2 / 3, k=k, nu_coarse=4, gram_schmidt=gs)
        p.sync(); t0 = time.perf_counter(); n = 10
        for _ in range(n): p.vcycle(4, 4, _lib.WJACOBI, omega=2 / 3, k=k, nu_coarse=4, gram_schmidt=gs)
        p.sync(); row["cycle_ms_gs_%s" % gs] = round((time.perf_counter() - t0) / n * 1e3, 3)
    p.sync(); t0 = time.perf_counter()
    for _ in range(10): p.gramschmidt(0, _lib.SLOT_V, k, modified=1)
    p.sync(); row["mgs_fine_ms"] = round((time.perf_counter() - t0) / 10 * 1e3, 3)
    print(json.dumps(row), flush=True)
    p.close()
