"""Counterpart of the reference's 1DPotMatrixVcycle.py (compute section :14-86, no plots): ten eigenpairs of the
1-D box, guesses from a 16-point Lanczos run, ten shift-and-invert steps whose solves are single V-cycles."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigridcmt_amd import drivers  # noqa: E402

out = drivers.shift_invert_eigenpairs("1d", gridsize=2 ** 7, bad_gridsize=2 ** 4, num_eigenvalues=10, max_iters=10,
                                      lowest_level=2 ** 4, tolerance=1e-4)
print("Initial guess eigenvalues: ", out["guess_eigenvalues"])
print("Eigenvalues Exact: ", [i ** 2 for i in range(1, 11)])
print("Eigenvalues Shift: ", out["eigenvalues"])
print("Discrete operator: ", drivers.exact_box_eigenvalues(2 ** 7, "1d", 10))
