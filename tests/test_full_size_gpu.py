"""GPU-only checks at BASELINE.json's sizes.

Direct parity: one V(2,2) cycle of the HIP path against the C oracle (oracle/mgcmt_oracle.c, the matrix-free
restatement of MGCMTSolver.py:281-329 pinned by the reference's golden vectors) on the same seeded input at the
configurations BASELINE.json names — 4096^2 red-black (config 2), 16384^2 weighted Jacobi and red-black (config 3),
32768^2 red-black on one GPU (config 4's workload), the 8192^2 square-well Hamiltonian (config 5).  The oracle
runs these on the box's host cores in about 0.2 / 3 / 12 / 1 s per cycle.  Plus size-independent properties: a
closed-form check (a discrete sine mode is an eigenvector of the Laplacian, so nu Jacobi sweeps scale it by a
known factor), fused == unfused kernels, linearity of the smoother."""
import numpy as np
import pytest

from conftest import rel_err
from multigridcmt_amd import _lib
from multigridcmt_amd.operators import laplacian_operator
from multigridcmt_amd.plan import Plan
from oracle import structured as st

pytestmark = pytest.mark.gpu
SCALE = -1 / np.pi ** 2


def _oracle_threads():
    import os
    st.lib().mgo_set_threads(min(os.cpu_count() or 1, 16))


def _one_cycle_vs_oracle(g, kind, omega, nu_coarse=2, seed=11, cycles=1, shift=0.0):
    """rel_err(HIP, oracle) of `cycles` V(2,2) cycles from a zero start on -laplacian(g)/pi^2 - shift."""
    _oracle_threads()
    f = np.random.RandomState(seed).rand(g * g)
    p = Plan(laplacian_operator(g, "2d") * SCALE, 8, nvec=1)
    try:
        p.set_shifts([shift])
        p.upload(0, _lib.SLOT_F, 0, f)
        p.fill(0, _lib.SLOT_V, 0, 0.0)
        for _ in range(cycles):
            p.vcycle(2, 2, kind, omega=omega, nu_coarse=nu_coarse)
        got = p.download(0, _lib.SLOT_V, 0)
    finally:
        p.close()
    X, Y = st.laplacian_factors(g, "2d", SCALE)
    v = np.zeros(g * g)
    okind = st.WJACOBI if kind == _lib.WJACOBI else st.GS_MC
    for _ in range(cycles):
        st.vcycle(X, Y, g, 8, shift, okind, v, f, 2, 2, nu_coarse, omega, inplace=True)
    del f
    return rel_err(got, v)


def test_config2_4096_redblack_cycle_against_oracle(hip_only):
    """BASELINE config 2: 4096^2, V(2,2) red-black, two cycles (the second one replays the captured graph), also with
    the reference's own V(4,4)-below-the-top quirk (MGCMTSolver.py:320) and a shift."""
    assert _one_cycle_vs_oracle(4096, _lib.GS_MC, 1.0, cycles=2) < 1e-10
    assert _one_cycle_vs_oracle(4096, _lib.GS_MC, 1.0, nu_coarse=4, shift=0.7) < 1e-10
    assert _one_cycle_vs_oracle(4096, _lib.WJACOBI, 2. / 3., nu_coarse=4) < 1e-10


@pytest.mark.parametrize("kind,omega", [(_lib.WJACOBI, 2. / 3.), (_lib.GS_MC, 1.0)])
def test_config3_16384_cycle_against_oracle(hip_only, kind, omega):
    """BASELINE config 3 (the headline workload): one V(2,2) cycle at 16384^2 through every optimised path."""
    assert _one_cycle_vs_oracle(16384, kind, omega) < 1e-10


def test_config4_32768_redblack_cycle_against_oracle(hip_only):
    """BASELINE config 4's workload on one GPU (the 8-GPU run shards exactly this cycle): 32768^2, V(2,2)
    red-black.  8 GiB per vector; the oracle needs about 45 GiB of host memory and 10-15 s."""
    assert _one_cycle_vs_oracle(32768, _lib.GS_MC, 1.0) < 1e-10


def test_config5_8192_square_well_against_oracle(hip_only):
    """BASELINE config 5 at its size: the 8192^2 square-well Hamiltonian (three Kronecker terms: Op5V on the finest
    level, Op9<3> below).  One V(2,2) cycle of each smoother on the eigen-residual of a random start, and the
    Rayleigh quotient of one iteration of drivers.potential_well_eigensolve, against the C oracle on the same
    operator."""
    import scipy.linalg
    from multigridcmt_amd.operators import potential_well_operator
    _oracle_threads()
    g = 8192
    op = potential_well_operator(g, 50.0, (g // 4, 3 * g // 4))
    X = np.ascontiguousarray(np.stack([t[0] for t in op.terms]))
    Y = np.ascontiguousarray(np.stack([t[1] for t in op.terms]))
    x = np.random.RandomState(0).random_sample(g * g)
    x /= np.linalg.norm(x)
    ax = st.apply(X, Y, 0.0, x)
    rho = float(np.dot(x, ax))
    r = ax - rho * x
    p = Plan(op, 8, nvec=1)
    try:
        p.set_shifts([0.0])
        p.upload(0, _lib.SLOT_W, 0, x)
        p.apply(0, (_lib.SLOT_W, 0), (_lib.SLOT_T, 0))
        assert rel_err(p.download(0, _lib.SLOT_T, 0), ax) < 1e-12
        p.upload(0, _lib.SLOT_F, 0, r)
        for kind, okind, omega in ((_lib.GS_MC, st.GS_MC, 1.0), (_lib.WJACOBI, st.WJACOBI, 2. / 3.)):
            p.fill(0, _lib.SLOT_V, 0, 0.0)
            p.vcycle(2, 2, kind, omega=omega, nu_coarse=2)
            w = p.download(0, _lib.SLOT_V, 0)
            w_ref = st.vcycle(X, Y, g, 8, 0.0, okind, np.zeros(g * g), r, 2, 2, 2, omega)
            assert rel_err(w, w_ref) < 1e-10
            if kind == _lib.GS_MC:
                # the Rayleigh-Ritz step of the driver on span{x, w}: rho of the new iterate
                def ritz(wv):
                    aw = st.apply(X, Y, 0.0, wv)
                    A2 = np.array([[rho, np.dot(x, aw)], [np.dot(x, aw), np.dot(wv, aw)]])
                    M2 = np.array([[1.0, np.dot(x, wv)], [np.dot(x, wv), np.dot(wv, wv)]])
                    return float(scipy.linalg.eigh(A2, M2)[0][0])
                want = ritz(w_ref)
                del w, w_ref
    finally:
        p.close()
    from multigridcmt_amd.drivers import potential_well_eigensolve
    hist = []
    potential_well_eigensolve(g, depth=50.0, cycles=1, method="vcycle", nu=2, lowest=8, smoother="rb", seed=0, history=hist)
    assert abs(hist[0] - want) < 1e-10 * abs(want)


def test_config3_16384_sine_mode_closed_form(hip_only):
    """Independent of every oracle: v_ij = sin(k pi (i+1)/(g+1)) sin(l pi (j+1)/(g+1)) is an eigenvector of the
    Dirichlet 5-point Laplacian, so with f = 0 nu weighted-Jacobi sweeps multiply it by (1 - omega lambda/d)^nu,
    lambda/d = sin^2(k pi/(2(g+1))) + sin^2(l pi/(2(g+1))); checked at the headline size for a smooth and an
    oscillatory mode, through the fused pass and for the operator application itself."""
    g = 16384
    i = np.arange(1, g + 1, dtype=np.int64)
    diag = abs(SCALE) * 4.0 * g * g

    def mode(k):                                          # sin(k pi i / (g+1)) with the argument reduced in integers
        return np.sin(np.pi * ((k * i) % (2 * (g + 1))) / (g + 1))

    p = Plan(laplacian_operator(g, "2d") * SCALE, g, nvec=1)
    try:
        p.set_shifts([0.0])
        p.fill(0, _lib.SLOT_F, 0, 0.0)
        for (k, l), nu in (((1, 2), 4), ((g // 2 + 3, g // 3 + 1), 3)):
            v = np.outer(mode(k), mode(l)).ravel()
            mu = np.sin(k * np.pi / (2 * (g + 1))) ** 2 + np.sin(l * np.pi / (2 * (g + 1))) ** 2
            lam = diag * mu                                        # eigenvalue of SCALE * laplacian (SCALE < 0)
            factor = (1.0 - (2. / 3.) * mu) ** nu                  # D = diag, so omega * lam / D = omega * mu
            nv = np.linalg.norm(v)
            p.upload(0, _lib.SLOT_V, 0, v)
            p.apply(0, (_lib.SLOT_V, 0), (_lib.SLOT_W, 0))
            # measured against |D| |v|: A v is a difference of terms of that size (for the smooth mode lam / D ~ 1e-7)
            assert np.linalg.norm(p.download(0, _lib.SLOT_W, 0) - lam * v) / (diag * nv) < 1e-13
            p.smooth(0, _lib.WJACOBI, nu, 2. / 3., k=1)
            assert rel_err(p.download(0, _lib.SLOT_V, 0), factor * v) < 1e-12
            del v
    finally:
        p.close()


def _residual_norm(p, n):
    p.apply(0, (_lib.SLOT_V, 0), (_lib.SLOT_W, 0), with_shift=True)
    p.axpy(0, -1.0, (_lib.SLOT_F, 0), (_lib.SLOT_W, 0))
    return np.sqrt(p.dot(0, (_lib.SLOT_W, 0), (_lib.SLOT_W, 0)))


@pytest.mark.parametrize("kind,omega", [(_lib.WJACOBI, 2. / 3.), (_lib.GS_MC, 1.0)])
def test_config2_4096_fused_equals_unfused_and_converges(hip_only, kind, omega):
    g = 4096
    f = np.random.RandomState(1).rand(g * g)
    outs, hist = [], []
    for fused in (1, 0):
        p = Plan(laplacian_operator(g, "2d") * SCALE, 8, nvec=1)
        p.set_option(_lib.OPT_FUSED, fused)
        p.set_shifts([0.0])
        p.upload(0, _lib.SLOT_F, 0, f)
        p.fill(0, _lib.SLOT_V, 0, 0.0)
        r = [np.linalg.norm(f)]
        for _ in range(4):
            p.vcycle(2, 2, kind, omega=omega, nu_coarse=2)
            r.append(_residual_norm(p, g * g))
        outs.append(p.download(0, _lib.SLOT_V, 0))
        hist.append(r)
        p.close()
    assert rel_err(outs[0], outs[1]) < 1e-10
    factors = [hist[0][i + 1] / hist[0][i] for i in range(4)]
    assert all(fac < 0.35 for fac in factors), factors            # V(2,2) contracts the residual every cycle
    # the same cycle on a grid the C oracle finishes in seconds reduces the residual at the same rate
    go = 1024
    X, Y = st.laplacian_factors(go, "2d", SCALE)
    fo = np.random.RandomState(1).rand(go * go)
    v = np.zeros(go * go)
    ro = [np.linalg.norm(fo)]
    for _ in range(4):
        v = st.vcycle(X, Y, go, 8, 0.0, st.WJACOBI if kind == _lib.WJACOBI else st.GS_MC, v, fo, 2, 2, 2, omega)
        ro.append(np.linalg.norm(st.residual(X, Y, 0.0, v, fo)))
    assert abs(ro[4] / ro[3] - factors[3]) < 0.05


@pytest.mark.parametrize("kind,omega", [(_lib.WJACOBI, 2. / 3.), (_lib.GS_MC, 1.0)])
def test_config3_16384_cycle_fused_equals_unfused(hip_only, kind, omega):
    """The headline configuration itself: one V(2,2) cycle at 16384^2 through every optimised path (fused passes,
    recompute instead of store, in-place colour windows, the LDS tail, graph capture) against the same cycle through
    the one-launch-per-operation kernels."""
    g = 16384
    f = np.random.RandomState(6).rand(g * g)
    outs = []
    for fused in (1, 0):
        p = Plan(laplacian_operator(g, "2d") * SCALE, 8, nvec=1)
        p.set_option(_lib.OPT_FUSED, fused)
        p.set_shifts([0.0])
        p.upload(0, _lib.SLOT_F, 0, f)
        p.fill(0, _lib.SLOT_V, 0, 0.0)
        for _ in range(2):                       # the second call replays the captured graph
            p.vcycle(2, 2, kind, omega=omega, nu_coarse=2)
        outs.append(p.download(0, _lib.SLOT_V, 0))
        p.close()
    assert rel_err(outs[0], outs[1]) < 1e-10


def test_config3_16384_jacobi_linearity(hip_only):
    """wjacobi with f = 0 is linear: S(a x + b y) = a S(x) + b S(y), checked at the headline size."""
    g = 16384
    p = Plan(laplacian_operator(g, "2d") * SCALE, g, nvec=3)
    p.set_shifts([0.0, 0.0, 0.0])
    rng = np.random.RandomState(5)
    x, y = rng.rand(g * g), rng.rand(g * g)
    p.upload(0, _lib.SLOT_V, 0, x)
    p.upload(0, _lib.SLOT_V, 1, y)
    p.upload(0, _lib.SLOT_V, 2, 2.0 * x - 0.5 * y)
    for q in range(3):
        p.fill(0, _lib.SLOT_F, q, 0.0)
    p.smooth(0, _lib.WJACOBI, 4, 2. / 3., k=3)
    sx, sy, sz = (p.download(0, _lib.SLOT_V, q) for q in range(3))
    p.close()
    assert rel_err(sz, 2.0 * sx - 0.5 * sy) < 1e-13


def test_sharded_plan_single_rank_on_gpu(hip_only):
    """World size 1 through the real device path of the sharded driver: zero-copy torch views of plan memory,
    device-side gather/scatter copies, an RCCL all-reduce.  Runs in a child process because PyTorch must be
    imported BEFORE libmgcmt_hip.so is loaded (both bring a HIP runtime; the first one loaded serves both).
    (Multi-rank exchange is covered on CPU with gloo, tests/test_distributed.py; the 8-GPU run belongs to the
    round driver.)"""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sharded_world1_gpu.py")], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "SHARDED_WORLD1_OK" in out.stdout


@pytest.mark.parametrize("world,kind_name", [(2, "wjacobi"), (4, "rb")])
def test_sharded_ranks_share_one_gpu(hip_only, tmp_path, world, kind_name):
    """Several ranks on the box's ONE GPU (gloo group, halo rows staged through host memory): the HIP strip kernels
    with real neighbours on both sides, the gather / scatter around the coarse problem and the recompute passes.
    The result must be the single-plan cycle's.  (RCCL cannot put two ranks on one device; the device-to-device
    exchange itself is exercised at world size 1 above and by the round driver's multi-GPU run.)"""
    import os
    import socket
    import subprocess
    import sys
    from conftest import ROOT
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
    s.close()
    g = 2048
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "sharded_gloo_gpu.py"), str(r), str(world), port, str(g),
                               kind_name, str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = []
    for pr in procs:
        try:
            outs.append(pr.communicate(timeout=600)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for r, (pr, out) in enumerate(zip(procs, outs)):
        assert pr.returncode == 0 and "SHARDED_GLOO_GPU_OK" in out, "rank %d: %s" % (r, out[-3000:])
    got = np.concatenate([np.load(tmp_path / ("part%d.npy" % r)) for r in range(world)])
    res, strip_levels = np.load(tmp_path / "res.npy")
    assert strip_levels == 2
    kind, omega = (_lib.WJACOBI, 2. / 3.) if kind_name == "wjacobi" else (_lib.GS_MC, 1.0)
    rng = np.random.RandomState(9)
    f, v0 = rng.rand(g * g), rng.rand(g * g)
    p = Plan(laplacian_operator(g, "2d") * SCALE, 8, nvec=1)
    p.set_shifts([0.3])
    p.upload(0, _lib.SLOT_F, 0, f)
    p.upload(0, _lib.SLOT_V, 0, v0)
    for _ in range(2):
        p.vcycle(2, 2, kind, omega=omega, nu_coarse=2)
    want = p.download(0, _lib.SLOT_V, 0)
    p.apply(0, (_lib.SLOT_V, 0), (_lib.SLOT_T, 0), with_shift=True)
    p.axpy(0, -1.0, (_lib.SLOT_F, 0), (_lib.SLOT_T, 0))
    want_res = np.sqrt(p.dot(0, (_lib.SLOT_T, 0), (_lib.SLOT_T, 0)))
    p.close()
    assert rel_err(got, want) < 1e-12
    assert abs(res - want_res) < 1e-9 * want_res


@pytest.mark.parametrize("g,kind,omega,nu", [(1024, _lib.WJACOBI, 2. / 3., 2), (1024, _lib.GS_MC, 1.0, 3), (512, _lib.WJACOBI, 2. / 3., 1)])
def test_graph_replay_equals_eager(hip_only, g, kind, omega, nu):
    """mgcmt_vcycle replays a captured HIP graph from its second call on; results must be those of eager launches,
    also when the shift changes between cycles (coarsest-level factorisation redone outside the graph) and when
    the number of buffer swaps per cycle is odd (two graphs alternate)."""
    f = np.random.RandomState(3).rand(g * g)
    outs = []
    for graph in (1, 0):
        p = Plan(laplacian_operator(g, "2d") * SCALE, 8, nvec=1)
        p.set_option(_lib.OPT_GRAPH, graph)
        p.upload(0, _lib.SLOT_F, 0, f)
        p.fill(0, _lib.SLOT_V, 0, 0.0)
        for i in range(6):
            p.set_shifts([0.0 if i < 3 else 0.5])
            p.vcycle(nu, nu, kind, omega=omega, nu_coarse=nu)
        outs.append(p.download(0, _lib.SLOT_V, 0))
        p.close()
    assert np.array_equal(outs[0], outs[1])


def test_transfers_pageable_and_page_locked_round_trip(hip_only):
    """mgcmt_upload / mgcmt_download above the staging threshold: pageable arrays (ring of pinned chunks, several host
    threads) and arrays of the result pool (one DMA) carry the same bytes; a result fed back as the next call's v0 is
    bit-identical to the same cycle from a pageable copy."""
    import gc
    from multigridcmt_amd import hostmem
    g = 2048                                           # 32 MiB per vector: staged / pooled
    rng = np.random.RandomState(5)
    p = Plan(laplacian_operator(g, "2d") * (-1 / np.pi ** 2), 8, nvec=1)
    x = rng.rand(g * g)
    p.upload(0, _lib.SLOT_F, 0, x)                     # pageable -> device (staged)
    back = p.download(0, _lib.SLOT_F, 0)               # device -> pool buffer (one DMA)
    assert back.base is not None and hostmem.stats()["pinned_bytes"] >= back.nbytes
    assert np.array_equal(back, x)
    plain = np.empty(g * g)
    p.download_into(0, _lib.SLOT_F, 0, plain)          # device -> pageable (staged)
    assert np.array_equal(plain, x)
    p.upload(0, _lib.SLOT_V, 0, back)                  # pool buffer -> device (one DMA)
    assert np.array_equal(p.download(0, _lib.SLOT_V, 0), x)
    # odd sizes: a tail chunk shorter than the ring's chunk, an offset view of a pool buffer
    p.set_shifts([0.0])
    p.upload(0, _lib.SLOT_V, 0, np.zeros(g * g))
    p.vcycle(2, 2, _lib.WJACOBI, omega=2. / 3.)
    v1 = p.download(0, _lib.SLOT_V, 0)                 # pooled result
    for src in (v1, np.array(v1)):                     # fed back from the pool / from a pageable copy
        p.upload(0, _lib.SLOT_V, 0, src)
        p.vcycle(2, 2, _lib.WJACOBI, omega=2. / 3.)
        out = np.array(p.download(0, _lib.SLOT_V, 0))
        if src is v1:
            first = out
    assert np.array_equal(first, out)
    p.close()
    del back, v1
    gc.collect()
    assert hostmem.stats()["free_buffers"] >= 1
