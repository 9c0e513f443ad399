"""Drop-in module name of the reference (`from MGCMTStencilMaker import MGCMTStencilMaker`, e.g. 1DPotMatrixVcycle.py:2-4)."""
from multigridcmt_amd import MGCMTStencilMaker  # noqa: F401
