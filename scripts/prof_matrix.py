import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multigridcmt_amd import _lib
from multigridcmt_amd.operators import laplacian_operator
from multigridcmt_amd.plan import Plan
k, g = 10, 4096
p = Plan(laplacian_operator(g, "2d") * (-1 / np.pi ** 2), 8, nvec=k)
p.set_shifts(np.linspace(1.9, 9.8, k))
rng = np.random.RandomState(0)
for q in range(k):
    p.upload(0, _lib.SLOT_F, q, rng.rand(g * g)); p.fill(0, _lib.SLOT_V, q, 0.0)
for _ in range(4): p.vcycle(4, 4, _lib.WJACOBI, omega=2 / 3, k=k, nu_coarse=4, gram_schmidt=True)
p.sync(); p.close()
