"""CPU tests of the boundary and the host logic: every symbol declared in include/mgcmt_hip.h is exported
by libmgcmt_hip.so (loaded, not executed: no GPU here), the StencilMaker matrices equal the reference's,
and operator recognition maps the reference's sparse matrices to Kronecker factors."""
import ctypes
import os
import re

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import ROOT, load_golden
from multigridcmt_amd import MGCMTStencilMaker, _lib
from multigridcmt_amd.operators import StructuredOperator, UnrecognisedOperator, laplacian_operator, recognise


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mgcmt_hip.h")).read()
    return sorted(set(re.findall(r"\b(mgcmt_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(_lib.EXPORTED_SYMBOLS)


def test_hip_library_exports_every_declared_symbol():
    path = _lib.DEFAULT_LIBRARY
    assert os.path.exists(path), "libmgcmt_hip.so missing: run __graft_entry__.build()"
    lib = ctypes.CDLL(path)                      # loading needs no GPU
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    lib.mgcmt_abi_version.restype = ctypes.c_int
    assert lib.mgcmt_abi_version() == _lib.ABI_VERSION


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "_lib_path", None)
    monkeypatch.setattr(_lib, "DEFAULT_LIBRARY", "/nonexistent/libmgcmt_hip.so")
    with pytest.raises(_lib.MgcmtError):
        _lib.lib()


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "multigridcmt_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, fn)).read()
                assert "oracle" not in text.replace("the oracle's", ""), os.path.join(dirpath, fn)


def test_stencil_maker_matrices_equal_reference():
    sm, g = MGCMTStencilMaker(), load_golden("operators")
    assert np.array_equal(sm.restriction(16, 8).toarray(), g["R_16_8"])
    assert np.array_equal(sm.interpolation(8, 16).toarray(), g["P_8_16"])
    assert np.array_equal(sm.interpolation(4, 16).toarray(), g["P_4_16"])
    assert np.array_equal(sm.restriction(16, 4).toarray(), g["R_16_4"])
    assert np.array_equal(sm.interpolation(4, 8, dimension="2d").toarray(), g["P2d_4_8"])
    assert np.array_equal(sm.restriction(8, 4, dimension="2d").toarray(), g["R2d_8_4"])
    assert np.array_equal(sm.restriction(16, 4, dimension="2d").toarray(), g["R2d_16_4"])      # fixed 1/4 quirk
    assert np.array_equal(sm.interpolation(4, 16, dimension="2d").toarray(), g["P2d_4_16"])
    assert np.array_equal(sm.laplacian(8).toarray(), g["L1d_8"])
    assert np.array_equal(sm.laplacian(4, dimension="2d").toarray(), g["L2d_4"])
    assert sm.laplacian(8).format == "csc"


def test_stencil_maker_error_convention(capsys):
    sm = MGCMTStencilMaker()
    assert sm.interpolation(8, 4) is None and sm.interpolation(6, 16) is None and sm.interpolation(4, 12) is None
    assert sm.restriction(4, 8) is None and sm.restriction(12, 4) is None
    out = capsys.readouterr().out
    assert "New gridsize isn't bigger than old gridsize !" in out and "Old gridsize isn't a power of 2 !" in out


def test_recognise_reference_operators():
    sm = MGCMTStencilMaker()
    op = recognise((-1 / np.pi ** 2) * sm.laplacian(16))
    assert op.dimension == "1d" and op.g == 16
    assert np.allclose(op.tocsr().toarray(), ((-1 / np.pi ** 2) * sm.laplacian(16)).toarray(), rtol=0, atol=0)
    A2 = (-1 / np.pi ** 2) * sm.laplacian(8, dimension="2d") - 1.7 * sp.eye(64)
    op2 = recognise(A2, "2d")
    assert op2.dimension == "2d" and np.abs(op2.tocsr().toarray() - A2.toarray()).max() < 1e-13
    V = np.add.outer(np.arange(8.0), 2 * np.arange(8.0)).ravel()                  # separable potential
    op3 = recognise(sm.laplacian(8, dimension="2d") + sp.diags(V), "2d")
    assert np.abs(op3.tocsr().toarray() - (sm.laplacian(8, dimension="2d") + sp.diags(V)).toarray()).max() < 1e-12
    with pytest.raises(UnrecognisedOperator):
        recognise(sm.laplacian(8, dimension="2d") + sp.diags(np.random.RandomState(0).rand(64)), "2d")
    with pytest.raises(UnrecognisedOperator):
        recognise(sp.random(64, 64, density=0.2, random_state=0) + sp.eye(64), "2d")


def test_recognise_sees_in_place_mutation():
    """A scipy matrix changed in place (same id, shape, nnz) must not come back as the stale cached operator."""
    sm = MGCMTStencilMaker()
    A = sp.csr_matrix(sm.laplacian(16))
    first = recognise(A)
    assert recognise(A) is first                                   # unchanged matrix: cache hit
    A *= 2.0
    second = recognise(A)
    assert second is not first and np.array_equal(second.diagonal(), A.diagonal())
    A.data[:] = 0.5 * A.data
    assert np.array_equal(recognise(A).diagonal(), A.diagonal())
    A.setdiag(-7.0)
    assert np.array_equal(recognise(A).diagonal(), A.diagonal())


def test_structured_operator_algebra():
    op = laplacian_operator(8, "2d")
    sm = MGCMTStencilMaker()
    assert np.array_equal(op.tocsr().toarray(), sm.laplacian(8, dimension="2d").toarray())
    assert np.allclose(((-0.5) * op).toarray(), -0.5 * sm.laplacian(8, dimension="2d").toarray())
    assert np.allclose(op.shifted(2.0).toarray(), (sm.laplacian(8, dimension="2d") - 2.0 * sp.eye(64)).toarray())
    assert np.allclose(op.diagonal(), sm.laplacian(8, dimension="2d").diagonal())
    assert isinstance(sm.laplacian(8, dimension="2d", matrix_free=True), StructuredOperator)


def test_bench_line_contract(capsys, monkeypatch):
    """bench.py's one JSON line carries every field of the driver's contract (run here on a tiny grid through the
    emulated kernels: the numbers mean nothing, the shape of the line is what is checked)."""
    import json
    import sys
    from conftest import ROOT, bind_backend
    bind_backend("emu")
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py", "--grid", "128", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    bench.main()
    line = [l for l in capsys.readouterr().out.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in out, key
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["warmup"] == 1 and out["dtype"] == "f64" and out["vs_baseline"] is None
    assert out["unit"] == "MLUPS" and out["higher_is_better"] is True and "workload" in out["config"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in out["roofline"], key
    assert out["roofline"]["bound"] == "hbm" and out["roofline"]["peak"] == 8000.0
    assert abs(out["roofline"]["frac"] - out["roofline"]["achieved"] / out["roofline"]["peak"]) < 1e-12
    assert len(out["residual_reduction_per_cycle"]) == 10 and out["residual_reduction_per_cycle"][-1] < 1e-3
    assert out["cycle_mehrstellen"]["four_colour"]["ms_per_step"] > 0 and out["cycle_mehrstellen"]["wjacobi"]["value"] > 0
    assert out["cycle_gauss_seidel_lexicographic"]["V22_ms_per_step"] > 0 and out["cycle_gauss_seidel_lexicographic"]["speedup_V22"] > 0


def test_lexwave_isa_keeps_load_destinations_in_place():
    """The lex pipeline's hand-counted asm loads (kernels_lexwave.hip) are only sound while their destination registers
    are never copied, spilled or reused as addresses inside the row loop: audited on the gfx950 ISA hipcc emits."""
    import os, shutil, subprocess, sys
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "audit_lexwave_isa.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_result_arrays_recycle_their_page_locked_buffers(backend, monkeypatch):
    """hostmem.empty: an ordinary ndarray over a buffer of mgcmt_host_alloc; the buffer returns to the free list when the
    last view dies; the limit and the size threshold fall back to numpy.empty."""
    import gc
    from multigridcmt_amd import hostmem
    hostmem.drain()
    base = hostmem.stats()
    n = (hostmem.MIN_BYTES // 8) + 5
    a = hostmem.empty(n)
    assert isinstance(a, np.ndarray) and a.dtype == np.float64 and a.size == n and a.flags.writeable and a.flags.c_contiguous
    a[:] = np.arange(n)
    address = a.ctypes.data
    view = a[3:9]
    del a
    gc.collect()
    assert hostmem.stats()["free_buffers"] == 0 and view[0] == 3.0          # a view keeps the buffer
    del view
    gc.collect()
    assert hostmem.stats()["free_buffers"] == 1
    b = hostmem.empty(n)
    assert b.ctypes.data == address and hostmem.stats()["reused"] == base["reused"] + 1
    small = hostmem.empty(16)
    assert small.base is None                                               # below the threshold: numpy's own memory
    monkeypatch.setenv("MGCMT_PINNED_POOL_BYTES", "0")
    assert hostmem.empty(n).base is None
    monkeypatch.setenv("MGCMT_PINNED_POOL_BYTES", str(8 * n + 8))           # room for exactly one buffer of this size
    c = hostmem.empty(n)                                                    # b holds it: fallback
    assert c.base is None
    del b
    gc.collect()
    d = hostmem.empty(2 * n)                                                # does not fit even after trimming
    assert d.base is None
    e = hostmem.empty(n)
    assert e.ctypes.data == address
    del e
    gc.collect()
    hostmem.drain()
    assert hostmem.stats()["pinned_bytes"] == 0
    lib = _lib.lib()
    assert lib.mgcmt_host_free(ctypes.c_void_p(12345)) != 0                  # not one of ours
    assert lib.mgcmt_host_alloc(0, ctypes.byref(ctypes.c_void_p())) != 0
