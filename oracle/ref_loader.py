"""Container-only tooling: run the Python-2 reference under Python 3 to make golden vectors.

TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Nothing under ``multigridcmt_amd/`` imports this.

The reference (``/root/reference/MGCMT{Solver,StencilMaker,Processor}.py``) is Python 2
(``print`` statements, ``xrange``, integer ``/``).  This loader reads the source *as text*,
applies the mechanical ``lib2to3`` fixers ``print``/``xrange``/``zip`` in memory, rewrites every
``/`` into a helper with Python-2 semantics (floor for int/int, true division otherwise; needed
at MGCMTSolver.py:81,394,397) and ``exec``s the result into a module registered under the
original name.  Nothing is written into the repository; the reference never travels to the GPU
box (``/root/reference`` does not exist there), so this module is only usable in the build
container, by ``oracle/gen_golden.py``.
"""
import ast
import os
import sys
import types
import warnings

import numpy as np

REFERENCE_ROOT = os.environ.get("MGCMT_REFERENCE_ROOT", "/root/reference")


def available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "MGCMTSolver.py"))


def _py2div(a, b):
    int_like = (int, np.integer)
    if isinstance(a, int_like) and not isinstance(a, bool) and isinstance(b, int_like):
        return a // b
    return a / b


class _DivRewriter(ast.NodeTransformer):
    def visit_BinOp(self, node):
        self.generic_visit(node)
        if isinstance(node.op, ast.Div):
            call = ast.Call(ast.Name("_py2div", ast.Load()), [node.left, node.right], [])
            return ast.copy_location(call, node)
        return node


def _load_one(name, path):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from lib2to3 import refactor
    tool = refactor.RefactoringTool(
        ["lib2to3.fixes.fix_print", "lib2to3.fixes.fix_xrange", "lib2to3.fixes.fix_zip"])
    with open(path) as fh:
        src = fh.read()
    if not src.endswith("\n"):
        src += "\n"
    tree = ast.parse(str(tool.refactor_string(src, path)), path)
    tree = _DivRewriter().visit(tree)
    ast.fix_missing_locations(tree)
    mod = types.ModuleType(name)
    mod.__file__ = path
    mod._py2div = _py2div
    sys.modules[name] = mod
    exec(compile(tree, path, "exec"), mod.__dict__)
    return mod


_CACHE = {}


def load_reference():
    """Returns (MGCMTSolver, MGCMTStencilMaker, MGCMTProcessor) classes of the reference."""
    if not _CACHE:
        if not available():
            raise RuntimeError("reference not present at %s" % REFERENCE_ROOT)
        saved = {m: sys.modules.get(m) for m in ("MGCMTStencilMaker", "MGCMTProcessor", "MGCMTSolver")}
        try:
            for m in ("MGCMTStencilMaker", "MGCMTProcessor", "MGCMTSolver"):
                _CACHE[m] = _load_one(m, os.path.join(REFERENCE_ROOT, m + ".py"))
        finally:
            # do not leave the reference registered under names the product shims also use
            for m, old in saved.items():
                if old is None:
                    sys.modules.pop(m, None)
                else:
                    sys.modules[m] = old
    return (_CACHE["MGCMTSolver"].MGCMTSolver,
            _CACHE["MGCMTStencilMaker"].MGCMTStencilMaker,
            _CACHE["MGCMTProcessor"].MGCMTProcessor)


def load_kp_model():
    """(PotWellSolver, Compound, GaAsValues, PotentialWell) of the reference's k.p model (PotWellSolver.py,
    baseCompounds.py, PotentialWell.py) for fixtures of SURVEY §8 (f)2.  Two mechanical shims, both for the interpreter
    and NumPy in this container rather than for the model: ``math`` is made visible (the module relied on
    ``from pylab import *`` leaking it, PotWellSolver.py:2,9) and the well boundaries, which the reference computes with
    np.floor / np.ceil and then uses as slice bounds (:28-32,41-43,151-152), are cast to int."""
    import builtins
    import math
    if "PotWellSolver" not in _CACHE:
        builtins.math = math
        load_reference()
        saved = {m: sys.modules.get(m) for m in ("baseCompounds", "PotentialWell", "PotWellSolver")}
        try:
            for m in ("baseCompounds", "PotentialWell", "PotWellSolver"):
                _CACHE[m] = _load_one(m, os.path.join(REFERENCE_ROOT, m + ".py"))
        finally:
            for m, old in saved.items():
                if old is None:
                    sys.modules.pop(m, None)
                else:
                    sys.modules[m] = old
        cls = _CACHE["PotWellSolver"].PotWellSolver
        original = cls.setParameters

        def set_parameters(self, *args, **kwargs):
            original(self, *args, **kwargs)
            self.potWellBoundary1 = int(self.potWellBoundary1)
            self.potWellBoundary2 = int(self.potWellBoundary2)

        cls.setParameters = set_parameters
    return (_CACHE["PotWellSolver"].PotWellSolver, _CACHE["baseCompounds"].Compound, _CACHE["baseCompounds"].GaAsValues,
            _CACHE["PotentialWell"].PotentialWell)
