"""The lexicographic Gauss-Seidel / SOR sweep as a pipeline of waves (csrc/kernels_lexwave.hip) against the C oracle's
sequential sweep (MGCMTSolver.py:210-246 restated in oracle/mgcmt_oracle.c) and against the one-workgroup kernel
(MGCMT_OPT_LEX_WAVE = 0), on the 5-point level and on 9-point Galerkin levels, with shifts, several vectors,
SOR incl. the reference's (D-L)^-1 right-hand-side quirk, and whole V-cycles."""
import numpy as np
import pytest

from conftest import rel_err
from multigridcmt_amd import _lib
from multigridcmt_amd.operators import laplacian_operator
from multigridcmt_amd.plan import Plan
from oracle import structured as st

SCALE = -1 / np.pi ** 2


def _galerkin_factors(g, levels):
    X, Y = st.laplacian_factors(g, "2d", SCALE)
    for _ in range(levels):
        X = np.stack([st.galerkin(t) for t in X])
        Y = np.stack([st.galerkin(t) for t in Y])
    return np.ascontiguousarray(X), np.ascontiguousarray(Y)


@pytest.mark.parametrize("g,level", [(128, 0), (256, 0), (256, 1), (512, 2), (64, 0), (32, 0), (16, 0), (128, 1), (128, 2), (128, 3)])
@pytest.mark.parametrize("kind,omega", [(_lib.GS_LEX, 1.0), (_lib.SOR_LEX, 1.3)])
def test_lex_wave_sweeps_match_oracle(backend, g, level, kind, omega):
    """nu sweeps on level `level` of a plan on grid g (level > 0: the 9-point Galerkin operator with its modified last
    row / column), two vectors with different shifts."""
    k, nu = 2, 3
    gl = g >> level
    rng = np.random.RandomState(7 + level)
    v0, f = rng.rand(k, gl * gl), rng.rand(k, gl * gl)
    shifts = [0.0, 0.8]
    outs = {}
    for wave in (2, 1, 0):
        p = Plan(laplacian_operator(g, "2d") * SCALE, 8, nvec=k)
        p.set_option(_lib.OPT_LEX_WAVE, wave)
        p.set_shifts(shifts)
        for q in range(k):
            p.upload(level, _lib.SLOT_V, q, v0[q])
            p.upload(level, _lib.SLOT_F, q, f[q])
        p.smooth(level, kind, nu, omega, k=k)
        outs[wave] = np.stack([p.download(level, _lib.SLOT_V, q) for q in range(k)])
        p.close()
    X, Y = _galerkin_factors(g, level)
    okind = st.GS_LEX if kind == _lib.GS_LEX else st.SOR_LEX
    for q in range(k):
        want = st.smooth(X, Y, shifts[q], okind, v0[q], f[q], nu, omega)
        assert rel_err(outs[2][q], want) < 1e-12, (q, "band wavefront vs oracle")
        assert rel_err(outs[1][q], want) < 1e-12, (q, "scan pipeline vs oracle")
        assert rel_err(outs[0][q], want) < 1e-12, (q, "one-workgroup kernel vs oracle")


@pytest.mark.parametrize("pipeline", [2, 1])
@pytest.mark.parametrize("kind,okind,omega", [(_lib.GS_LEX, st.GS_LEX, 1.0), (_lib.SOR_LEX, st.SOR_LEX, 1.2)])
def test_lex_wave_vcycle_matches_oracle(backend, kind, okind, omega, pipeline):
    """Whole V-cycles with the reference's default smoother (ThesisProblem.py:101 passes smoother=solver.gseidel):
    the levels of at least 16 x 16 points take the wave pipeline, the coarsest ones the one-workgroup kernel."""
    g = 512
    f = np.random.RandomState(3).rand(g * g)
    p = Plan(laplacian_operator(g, "2d") * SCALE, 8, nvec=1)
    p.set_option(_lib.OPT_LEX_WAVE, pipeline)
    p.set_shifts([0.4])
    p.upload(0, _lib.SLOT_F, 0, f)
    p.fill(0, _lib.SLOT_V, 0, 0.0)
    for _ in range(2):
        p.vcycle(2, 2, kind, omega=omega, nu_coarse=4)
    got = p.download(0, _lib.SLOT_V, 0)
    p.close()
    X, Y = st.laplacian_factors(g, "2d", SCALE)
    v = np.zeros(g * g)
    for _ in range(2):
        v = st.vcycle(X, Y, g, 8, 0.4, okind, v, f, 2, 2, 4, omega)
    assert rel_err(got, v) < 1e-11


@pytest.mark.gpu
@pytest.mark.parametrize("pipeline", [2, 1])
def test_lex_wave_4096_cycle_against_oracle(hip_only, pipeline):
    """The reference's default Gauss-Seidel at BASELINE config 2's grid: one V(2,2) cycle, every level through the
    wave pipeline down to 128^2 (several hundred waves in flight, hand-offs under load)."""
    g = 4096
    st.lib().mgo_set_threads(16)
    f = np.random.RandomState(8).rand(g * g)
    p = Plan(laplacian_operator(g, "2d") * SCALE, 8, nvec=1)
    p.set_option(_lib.OPT_LEX_WAVE, pipeline)
    p.set_shifts([0.0])
    p.upload(0, _lib.SLOT_F, 0, f)
    p.fill(0, _lib.SLOT_V, 0, 0.0)
    p.vcycle(2, 2, _lib.GS_LEX, omega=1.0, nu_coarse=2)
    got = p.download(0, _lib.SLOT_V, 0)
    p.close()
    X, Y = st.laplacian_factors(g, "2d", SCALE)
    want = st.vcycle(X, Y, g, 8, 0.0, st.GS_LEX, np.zeros(g * g), f, 2, 2, 2, 1.0)
    assert rel_err(got, want) < 1e-10


@pytest.mark.parametrize("g,level,nu", [(256, 0, 2), (256, 1, 4), (128, 0, 3), (64, 1, 2)])
def test_chained_sweeps_equal_separate_sweeps(backend, g, level, nu):
    """MGCMT_OPT_LEX_CHAIN: the nu Gauss-Seidel sweeps of a smoothing step in ONE launch (sweep s + 1 runs a few rows
    behind sweep s, reading its results as they become visible) — the same arithmetic in the same order, so the same
    bits as one launch per sweep; and both against the oracle."""
    gl = g >> level
    rng = np.random.RandomState(21)
    k = 2
    v0, f = rng.rand(k, gl * gl), rng.rand(k, gl * gl)
    shifts = [0.0, 0.3]
    outs = {}
    for chain in (1, 0):
        p = Plan(laplacian_operator(g, "2d") * SCALE, 8, nvec=k)
        p.set_option(_lib.OPT_LEX_WAVE, 1)
        p.set_option(_lib.OPT_LEX_CHAIN, chain)
        p.set_shifts(shifts)
        for q in range(k):
            p.upload(level, _lib.SLOT_V, q, v0[q])
            p.upload(level, _lib.SLOT_F, q, f[q])
        p.smooth(level, _lib.GS_LEX, nu, 1.0, k=k)
        outs[chain] = np.stack([p.download(level, _lib.SLOT_V, q) for q in range(k)])
        p.close()
    assert np.array_equal(outs[1], outs[0])
    X, Y = _galerkin_factors(g, level)
    for q in range(k):
        assert rel_err(outs[1][q], st.smooth(X, Y, shifts[q], st.GS_LEX, v0[q], f[q], nu, 1.0)) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("g,kind,omega", [(16384, _lib.GS_LEX, 1.0), (8192, _lib.SOR_LEX, 1.3)])
def test_chained_cycle_equals_unchained_at_size(hip_only, g, kind, omega):
    """Thousands of waves, hand-offs and chases under load: three V(2,2) cycles with the sweeps of every smoothing step
    chained in one launch give the bits of one launch per sweep (visibility of the sweep ahead's stores is the one thing
    the emulation cannot check)."""
    f = np.random.RandomState(12).rand(g * g)
    outs = []
    for chain in (1, 0):
        p = Plan(laplacian_operator(g, "2d") * SCALE, 8, nvec=1)
        p.set_option(_lib.OPT_LEX_CHAIN, chain)
        p.set_shifts([0.0])
        p.upload(0, _lib.SLOT_F, 0, f)
        p.fill(0, _lib.SLOT_V, 0, 0.0)
        for _ in range(3):
            p.vcycle(2, 2, kind, omega=omega, nu_coarse=4)
        outs.append(np.array(p.download(0, _lib.SLOT_V, 0)))
        p.close()
    assert np.array_equal(outs[0], outs[1])


def test_lex_scratch_growth_between_graph_replays(backend):
    """One plan, lexicographic Gauss-Seidel: V(2,2) three times (eager, capture, replay), then V(4,4) — whose chained
    sweeps need more pipeline scratch, so the buffers the captured V(2,2) graph points at are reallocated —, then V(2,2)
    again: the cached graphs must have been dropped with the old buffers (a replay through freed memory otherwise: a
    memory fault or silent corruption on the GPU).  Every step against the C oracle.  (On the emulation, where graph
    capture is a stub, this is the same sequence run eagerly.)"""
    g = 256 if backend == "hip" else 64
    f = np.random.RandomState(5).rand(g * g)
    p = Plan(laplacian_operator(g, "2d") * SCALE, 8, nvec=1)
    p.set_shifts([0.3])
    p.upload(0, _lib.SLOT_F, 0, f)
    p.fill(0, _lib.SLOT_V, 0, 0.0)
    X, Y = st.laplacian_factors(g, "2d", SCALE)
    v = np.zeros(g * g)
    for nu in (2, 2, 2, 4, 2, 2, 5, 2):
        p.vcycle(nu, nu, _lib.GS_LEX, omega=1.0, nu_coarse=nu)
        v = st.vcycle(X, Y, g, 8, 0.3, st.GS_LEX, v, f, nu, nu, nu, 1.0)
        got = p.download(0, _lib.SLOT_V, 0)          # (also the synchronising call that looks at the pipeline's error word)
        assert rel_err(got, v) < 1e-11, nu
    p.close()
