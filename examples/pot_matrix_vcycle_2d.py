"""Counterpart of the reference's 2DPotMatrixVcycle.py (compute section :15-120, no plots): ten eigenpairs of the
2-D box on 64 x 64, guesses from 16 x 16, five steps, coarsest V-cycle level 8."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigridcmt_amd import drivers  # noqa: E402

g = int(sys.argv[1]) if len(sys.argv) > 1 else 2 ** 6
out = drivers.shift_invert_eigenpairs("2d", gridsize=g, bad_gridsize=2 ** 4, num_eigenvalues=10, max_iters=5,
                                      lowest_level=2 ** 3, tolerance=np.finfo(float).eps, verbose=True)
nx = [1, 2, 1, 2, 3, 1, 2, 3, 1, 4]
ny = [1, 1, 2, 2, 1, 3, 3, 2, 4, 1]
print("Exact eigenvalues:", [a * a + b * b for a, b in zip(nx, ny)])
print("Multigrid eigenvalues:", out["eigenvalues"])
print("Discrete operator:", drivers.exact_box_eigenvalues(g, "2d", 10))
