"""Counterpart of the reference's RQMin.py (:15-50, no plots): Rayleigh-quotient multigrid for the two lowest states."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigridcmt_amd import drivers  # noqa: E402

start = time.perf_counter()
rho, rho2, _ = drivers.rayleigh_quotient_multigrid(2 ** 6, 2, 10)
print(rho)
print(rho2)
print("exact:", drivers.exact_box_eigenvalues(2 ** 6, "1d", 2))
print("RQMG time: ", time.perf_counter() - start)
