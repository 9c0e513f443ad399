"""GPU-only checks at BASELINE.json's sizes, through size-independent properties (the oracle cannot run
a 16384^2 cycle in seconds): fused == unfused kernels, linearity of the smoother, and per-cycle residual
reduction equal to what the C oracle shows at a size it can run."""
import numpy as np
import pytest

from conftest import rel_err
from multigridcmt_amd import _lib
from multigridcmt_amd.operators import laplacian_operator
from multigridcmt_amd.plan import Plan
from oracle import structured as st

pytestmark = pytest.mark.gpu
SCALE = -1 / np.pi ** 2


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
