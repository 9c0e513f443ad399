"""Drop-in module name of the reference (`from MGCMTSolver import MGCMTSolver`, e.g. 1DPotMatrixVcycle.py:2-4)."""
from multigridcmt_amd import MGCMTSolver  # noqa: F401
