"""Drop-in module name of the reference (`from MGCMTProcessor import MGCMTProcessor`, e.g. 1DPotMatrixVcycle.py:2-4)."""
from multigridcmt_amd import MGCMTProcessor  # noqa: F401
