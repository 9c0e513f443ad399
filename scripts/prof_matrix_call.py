import cProfile, io, os, pstats, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multigridcmt_amd import MGCMTSolver, MGCMTStencilMaker
g, k = 1024, 10
solver, sm = MGCMTSolver(), MGCMTStencilMaker()
A = sm.laplacian(g, dimension="2d", matrix_free=True) * (-1 / np.pi ** 2)
rng = np.random.RandomState(0)
F = rng.rand(g * g, k); W0 = np.zeros((g * g, k)); sh = np.linspace(1.9, 9.8, k)
solver.vcycle_matrix(W0, F, A, sm, shifts=sh, dimension="2d", lowest_level=8)
pr = cProfile.Profile(); pr.enable()
for _ in range(3): solver.vcycle_matrix(W0, F, A, sm, shifts=sh, dimension="2d", lowest_level=8)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(10); print(s.getvalue()[:2500])
