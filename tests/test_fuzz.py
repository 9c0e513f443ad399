"""Randomised A/B of the optimised cycle (fused passes, recompute, in-place colour windows, LDS tail, graph replay)
against the one-launch-per-operation kernels over grid sizes, coarsest levels, sweep counts, smoothers, relaxation
factors, shifts, vector counts and option settings.  A few cases on the emulator, many on the GPU."""
import os

import numpy as np
import pytest

from conftest import rel_err
from multigridcmt_amd import _lib
from multigridcmt_amd.operators import laplacian_operator, potential_well_operator
from multigridcmt_amd.plan import Plan


def _case(rng, sizes, dense_tail=False):
    g = int(rng.choice(sizes))
    lowest = int(rng.choice([s for s in (2, 4, 8, 16) if s < g]))
    kind = int(rng.choice([_lib.WJACOBI, _lib.GS_MC]))
    omega = float(rng.choice([2. / 3., 0.8])) if kind == _lib.WJACOBI else float(rng.choice([1.0, 1.15]))
    nu1, nu2, nuc = (int(x) for x in rng.randint(0, 4, size=3))
    k = int(rng.randint(1, 4))
    well = bool(rng.randint(0, 3) == 0)
    opts = {_lib.OPT_RECOMPUTE: int(rng.choice([0, 1, 2])), _lib.OPT_TAIL: int(rng.randint(0, 3) if dense_tail else rng.choice([0, 2])), _lib.OPT_GRAPH: int(rng.randint(0, 2))}
    zero_start = bool(rng.randint(0, 3) == 0)    # every cycle starts from "V is zero" as a flag (MGCMT_CYCLE_ZERO_START); V holds garbage
    return g, lowest, kind, omega, nu1, nu2, nuc, k, well, opts, zero_start


def _run_case(case, seed):
    g, lowest, kind, omega, nu1, nu2, nuc, k, well, opts, zero_start = case
    op = potential_well_operator(g, 20.0, (g // 4, 3 * g // 4)) if well else laplacian_operator(g, "2d") * (-1 / np.pi ** 2)
    rng = np.random.RandomState(seed)
    v0, f = rng.rand(k, g * g), rng.rand(k, g * g)
    outs = []
    for fused in (1, 0):
        p = Plan(op, lowest, nvec=k)
        p.set_option(_lib.OPT_FUSED, fused)
        for o, val in opts.items():
            p.set_option(o, val)
        p.set_shifts(0.2 + 0.5 * np.arange(k))
        for q in range(k):
            p.upload(0, _lib.SLOT_V, q, v0[q] * (1e6 if zero_start else 1.0))
            p.upload(0, _lib.SLOT_F, q, f[q])
        for _ in range(3):                      # with graph replay from the second call on
            p.vcycle(nu1, nu2, kind, omega=omega, k=k, nu_coarse=nuc, zero_start=zero_start)
        outs.append(np.stack([p.download(0, _lib.SLOT_V, q) for q in range(k)]))
        p.close()
    return rel_err(outs[0], outs[1])


def test_random_cycles_small(backend):
    rng = np.random.RandomState(int(os.environ.get("MGCMT_FUZZ_SEED", 2024 if backend == "emu" else 7)))
    n, sizes = (6, (32, 64, 128)) if backend == "emu" else (60, (32, 64, 128, 256, 512, 1024, 2048))
    n = int(os.environ.get("MGCMT_FUZZ_CASES", n))          # longer soak runs: MGCMT_FUZZ_CASES=500 MGCMT_FUZZ_SEED=...
    for i in range(n):
        case = _case(rng, sizes, dense_tail=backend == "hip")
        err = _run_case(case, 100 + i)
        assert err < 1e-11, (case, err)
