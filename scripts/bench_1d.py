"""1-D cycle at n = 2^24 (bench.py's cycle_1d leg alone).  usage: bench_1d.py [log2 n]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multigridcmt_amd import _lib
from multigridcmt_amd.operators import laplacian_operator
from multigridcmt_amd.plan import Plan
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 24)
p = Plan(laplacian_operator(n, "1d") * (-1.0 / np.pi ** 2), 8, nvec=1)
p.set_shifts([0.0])
p.upload(0, _lib.SLOT_F, 0, np.random.RandomState(1).rand(n))
out = {}
for name, kind, om in (("wjacobi", _lib.WJACOBI, 2 / 3), ("rb", _lib.GS_MC, 1.0)):
    for nu in (2, 4):
        p.fill(0, _lib.SLOT_V, 0, 0.0)
        for _ in range(5):
            p.vcycle(nu, nu, kind, omega=om, k=1, nu_coarse=nu)
        p.sync(); t0 = time.perf_counter()
        for _ in range(50):
            p.vcycle(nu, nu, kind, omega=om, k=1, nu_coarse=nu)
        p.sync(); out["V%d%d_%s_ms" % (nu, nu, name)] = round((time.perf_counter() - t0) / 50 * 1e3, 4)
    ms = p.time_smoother(0, kind, 2, om, 20) / 20
    out["pass_%s_ms" % name] = round(ms, 4); out["pass_%s_TBs_24B" % name] = round(n * 24 / ms / 1e9, 3)
print(json.dumps(out))
