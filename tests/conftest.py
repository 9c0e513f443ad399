import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "hip_cpu_mock")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def rel_err(a, b):
    a, b = np.asarray(a), np.asarray(b)
    kind = complex if (np.iscomplexobj(a) or np.iscomplexobj(b)) else float
    a, b = a.astype(kind), b.astype(kind)
    return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-300))


_bound = {"kind": None}


def bind_backend(kind):
    """Point the ctypes binding at the real HIP library ("hip") or at the host-only emulation build
    of the same kernel sources ("emu", tests/hip_cpu_mock).  Only tests do this."""
    from multigridcmt_amd import _lib, plan
    if _bound["kind"] == kind:
        return
    plan.release_plans()
    if kind == "hip":
        os.environ.pop("MGCMT_TAIL_DENSE", None)
        path = os.environ.get("MGCMT_TEST_LIBRARY", _lib.DEFAULT_LIBRARY)     # a tuning variant of the HIP library
        if not os.path.exists(path):
            pytest.fail("%s is not built — run __graft_entry__.build()" % path)
        _lib.use_library(path)
        if _lib.device_count() < 1:
            pytest.fail("no HIP device visible")
    else:
        import build_emu
        # the dense form of the cycle's tail forms its matrix with 1024 workgroups of 1024 threads: minutes on the emulation.
        # CPU tests run the LDS-resident tail unless a test asks for the dense form (tests/test_fused_kernels.py)
        os.environ["MGCMT_TAIL_DENSE"] = "0"
        _lib.use_library(build_emu.build())
    _bound["kind"] = kind


@pytest.fixture(params=[pytest.param("emu"), pytest.param("hip", marks=pytest.mark.gpu)])
def backend(request):
    """Every parity test runs twice: through the HIP library on the GPU box (-m gpu) and through the
    emulated kernels on CPU (-m "not gpu") so kernel logic regressions show up without a GPU."""
    bind_backend(request.param)
    return request.param


@pytest.fixture
def hip_only():
    bind_backend("hip")
    return "hip"
