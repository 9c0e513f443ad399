"""2DPotMatrixVcycle.py's outer loop (:54-109) with all vectors resident on the GPU: the ten lowest eigenpairs of
-laplacian/pi^2 on a box, guesses from a 16^2 grid.  usage: pot_matrix_vcycle_2d_resident.py [gridsize] [iterations]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigridcmt_amd import drivers  # noqa: E402

g = int(sys.argv[1]) if len(sys.argv) > 1 else 2 ** 10
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
stats = {}
start = time.perf_counter()
out = drivers.shift_invert_eigenpairs_resident("2d", g, 16, 10, iters, lowest_level=8, tolerance=np.finfo(float).eps, stats=stats)
print("grid %d^2, 10 eigenpairs, %d iterations: %.3f s in the loop (%.1f ms per iteration), %.2f s in total" %
      (g, iters, stats["loop_seconds"], stats["loop_seconds"] / iters * 1e3, time.perf_counter() - start))
print("multigrid eigenvalues:", np.sort(out["eigenvalues"]))
print("exact (discrete)     :", drivers.exact_box_eigenvalues(g, "2d", 10))
