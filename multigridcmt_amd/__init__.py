"""multigridcmt_amd — the geometric-multigrid V-cycle path of AndyMN/MultigridCMT on AMD MI355X.

Host side: the reference's three classes with unchanged signatures.  Device side: hand-written HIP
kernels for gfx950 behind the C-ABI of include/mgcmt_hip.h (libmgcmt_hip.so, bound with ctypes).
"""
from .operators import (StructuredOperator, UnrecognisedOperator, identity_operator, laplacian_operator,
                        potential_well_operator, recognise)
from .plan import Plan, get_plan, release_plans
from .processor import MGCMTProcessor
from .solver import MGCMTSolver
from .stencil_maker import MGCMTStencilMaker

__all__ = ["MGCMTSolver", "MGCMTStencilMaker", "MGCMTProcessor", "StructuredOperator", "UnrecognisedOperator",
           "laplacian_operator", "identity_operator", "potential_well_operator", "recognise", "Plan", "get_plan", "release_plans"]
