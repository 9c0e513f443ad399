import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multigridcmt_amd import drivers
g = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
drivers.potential_well_eigensolve(g, cycles=2, method="vcycle")
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
drivers.potential_well_eigensolve(g, cycles=10, method="vcycle")
print("10 cycles:", time.perf_counter() - t0)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18); print(s.getvalue()[:4000])
