"""The sharded V-cycle (mgcmt_sharded_vcycle of csrc/sharded.hip behind multigridcmt_amd/distributed.py) with
world_size 2 and 4 on CPU: the library's external transport carried by gloo, the emulated kernels for the compute
(boundary rows first, then the interior rows, as on the GPU).  The sharded result must equal the single-plan result."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, rel_err


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _operator(name, g):
    from multigridcmt_amd.operators import laplacian_operator, potential_well_operator
    if name == "well":        # BASELINE config 5's Hamiltonian: 5-point + product potential on the strips of the finest level,
        return potential_well_operator(g, 40.0, (g // 4, 3 * g // 4))     # constant part + one variable term below it
    return laplacian_operator(g, "2d") * (-1 / np.pi ** 2)


def _worker(rank, world, port, g, kind, omega, nu, out_dir, force_recompute, opname="laplacian"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "hip_cpu_mock")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import build_emu
    from multigridcmt_amd import _lib
    _lib.use_library(build_emu.build())
    from multigridcmt_amd.distributed import ShardedPlan
    dist.init_process_group("gloo", rank=rank, world_size=world)
    op = _operator(opname, g)
    sp = ShardedPlan(op, 8, rank, world, switch_grid=g // 4, on_gpu=False)
    sp.set_shift(0.4)
    sp.set_comm_option(_lib.COMM_OPT_SPLIT, 2)      # boundary rows first on these small strips too
    if force_recompute:
        sp.plan.set_option(_lib.OPT_RECOMPUTE, 2)   # take the recompute-instead-of-store passes on these small strips too
    rng = np.random.RandomState(5)
    f, v0 = rng.rand(g * g), rng.rand(g * g)
    rows = g // world
    sl = slice(rank * rows * g, (rank + 1) * rows * g)
    sp.upload_local(_lib.SLOT_F, f[sl])
    sp.upload_local(_lib.SLOT_V, v0[sl])
    for _ in range(2):
        sp.vcycle(nu, nu, kind, omega=omega, nu_coarse=nu)
    res = sp.residual_norm()
    np.save(os.path.join(out_dir, "part%d.npy" % rank), sp.download_local(_lib.SLOT_V))
    if rank == 0:
        np.save(os.path.join(out_dir, "res.npy"), np.array([res, sp.strip_levels]))
    sp.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,force_recompute,g", [(2, False, 512), (4, False, 1024), (2, True, 512)])
@pytest.mark.parametrize("kind_name", ["wjacobi", "rb"])
def test_sharded_cycle_equals_single_plan(tmp_path, world, kind_name, force_recompute, g):
    import torch.multiprocessing as mp
    from conftest import bind_backend
    bind_backend("emu")
    from multigridcmt_amd import _lib
    from multigridcmt_amd.operators import laplacian_operator
    from multigridcmt_amd.plan import Plan
    nu = 2
    kind, omega = (_lib.WJACOBI, 2. / 3.) if kind_name == "wjacobi" else (_lib.GS_MC, 1.0)
    mp.spawn(_worker, args=(world, _free_port(), g, kind, omega, nu, str(tmp_path), force_recompute), nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / ("part%d.npy" % r)) for r in range(world)])
    res, strip_levels = np.load(tmp_path / "res.npy")
    assert strip_levels == 2
    p = Plan(laplacian_operator(g, "2d") * (-1 / np.pi ** 2), 8, nvec=1)
    p.set_shifts([0.4])
    rng = np.random.RandomState(5)
    f, v0 = rng.rand(g * g), rng.rand(g * g)
    p.upload(0, _lib.SLOT_F, 0, f)
    p.upload(0, _lib.SLOT_V, 0, v0)
    for _ in range(2):
        p.vcycle(nu, nu, kind, omega=omega, nu_coarse=nu)
    want = p.download(0, _lib.SLOT_V, 0)
    p.apply(0, (_lib.SLOT_V, 0), (_lib.SLOT_T, 0), with_shift=True)
    p.axpy(0, -1.0, (_lib.SLOT_F, 0), (_lib.SLOT_T, 0))
    want_res = np.sqrt(p.dot(0, (_lib.SLOT_T, 0), (_lib.SLOT_T, 0)))
    p.close()
    assert rel_err(got, want) < 1e-12
    assert abs(res - want_res) < 1e-9 * want_res


def test_sharded_cycle_with_eight_ranks(tmp_path):
    """The job size the round driver launches (N = 8): two end ranks, six interior ones, eight strips gathered for the
    redundant coarse sub-cycle — 1024^2, strips of 128 rows down to 256^2, against the single plan."""
    test_sharded_cycle_equals_single_plan(tmp_path, 8, "rb", False, 1024)


@pytest.mark.parametrize("kind_name", ["wjacobi", "rb"])
def test_sharded_square_well_equals_single_plan(tmp_path, kind_name):
    """The same on the square-well Hamiltonian (BASELINE config 5's operator): the strips of the finest level run the
    5-point-plus-product-potential passes, the strips below it the constant-part-plus-variable-term passes, whose row
    factors carry halo entries of the neighbours' rows; the residual norm goes through the marching operator
    application on a strip."""
    import torch.multiprocessing as mp
    from conftest import bind_backend
    bind_backend("emu")
    from multigridcmt_amd import _lib
    from multigridcmt_amd.plan import Plan
    g, nu, world = 256, 2, 2
    kind, omega = (_lib.WJACOBI, 2. / 3.) if kind_name == "wjacobi" else (_lib.GS_MC, 1.0)
    mp.spawn(_worker, args=(world, _free_port(), g, kind, omega, nu, str(tmp_path), False, "well"), nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / ("part%d.npy" % r)) for r in range(world)])
    res, strip_levels = np.load(tmp_path / "res.npy")
    assert strip_levels == 2
    p = Plan(_operator("well", g), 8, nvec=1)
    assert [p.operator_kind(l) for l in range(2)] == [_lib.OPK_FIVE_DIAG, _lib.OPK_NINE_VAR]
    p.set_shifts([0.4])
    rng = np.random.RandomState(5)
    f, v0 = rng.rand(g * g), rng.rand(g * g)
    p.upload(0, _lib.SLOT_F, 0, f)
    p.upload(0, _lib.SLOT_V, 0, v0)
    for _ in range(2):
        p.vcycle(nu, nu, kind, omega=omega, nu_coarse=nu)
    want = p.download(0, _lib.SLOT_V, 0)
    p.apply(0, (_lib.SLOT_V, 0), (_lib.SLOT_T, 0), with_shift=True)
    p.axpy(0, -1.0, (_lib.SLOT_F, 0), (_lib.SLOT_T, 0))
    want_res = np.sqrt(p.dot(0, (_lib.SLOT_T, 0), (_lib.SLOT_T, 0)))
    p.close()
    assert rel_err(got, want) < 1e-12
    assert abs(res - want_res) < 1e-9 * want_res


def _matrix_worker(rank, world, port, g, kind, omega, nu, k, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "hip_cpu_mock")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import build_emu
    from multigridcmt_amd import _lib
    _lib.use_library(build_emu.build())
    from multigridcmt_amd.distributed import ShardedPlan
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sp = ShardedPlan(_operator("laplacian", g), 8, rank, world, switch_grid=g // 4, on_gpu=False, nvec=k)
    sp.set_shifts(np.linspace(0.1, 0.5, k))
    sp.set_comm_option(_lib.COMM_OPT_SPLIT, 2)
    rng = np.random.RandomState(11)
    F, V0 = rng.rand(k, g * g), rng.rand(k, g * g)
    rows = g // world
    sl = slice(rank * rows * g, (rank + 1) * rows * g)
    for q in range(k):
        sp.upload_local(_lib.SLOT_F, F[q, sl], vec=q)
        sp.upload_local(_lib.SLOT_V, V0[q, sl], vec=q)
    for _ in range(2):
        sp.vcycle(nu, nu, kind, omega=omega, nu_coarse=nu, k=k, gram_schmidt=True)
    np.save(os.path.join(out_dir, "part%d.npy" % rank), np.stack([sp.download_local(_lib.SLOT_V, vec=q) for q in range(k)]))
    sp.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind_name", ["wjacobi", "rb"])
def test_sharded_vcycle_matrix_equals_single_plan(tmp_path, kind_name):
    """vcycle_matrix (MGCMTSolver.py:375-436) on strips: k columns with their own shifts, modified Gram-Schmidt on every
    level on the way up (:434) with the inner products all-reduced over the ranks (one all-reduce of the column's
    coefficients per column) — against the single plan's cycle with MGCMT_CYCLE_GRAM_SCHMIDT."""
    import torch.multiprocessing as mp
    from conftest import bind_backend
    bind_backend("emu")
    from multigridcmt_amd import _lib
    from multigridcmt_amd.plan import Plan
    g, nu, k, world = 256, 2, 3, 2
    kind, omega = (_lib.WJACOBI, 2. / 3.) if kind_name == "wjacobi" else (_lib.GS_MC, 1.0)
    mp.spawn(_matrix_worker, args=(world, _free_port(), g, kind, omega, nu, k, str(tmp_path)), nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / ("part%d.npy" % r)) for r in range(world)], axis=1)
    p = Plan(_operator("laplacian", g), 8, nvec=k)
    p.set_shifts(np.linspace(0.1, 0.5, k))
    rng = np.random.RandomState(11)
    F, V0 = rng.rand(k, g * g), rng.rand(k, g * g)
    for q in range(k):
        p.upload(0, _lib.SLOT_F, q, F[q])
        p.upload(0, _lib.SLOT_V, q, V0[q])
    for _ in range(2):
        p.vcycle(nu, nu, kind, omega=omega, k=k, nu_coarse=nu, gram_schmidt=True)
    want = np.stack([p.download(0, _lib.SLOT_V, q) for q in range(k)])
    p.close()
    assert rel_err(got, want) < 1e-12
    gram = want @ want.T
    assert np.abs(gram - np.eye(k)).max() < 1e-12          # (orthonormal columns: what the Gram-Schmidt is for)


def _bench_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "hip_cpu_mock")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import argparse
    import contextlib
    import io
    import build_emu
    from multigridcmt_amd import _lib, dist_bench
    _lib.use_library(build_emu.build())
    args = argparse.Namespace(grid=512, smoother="rb", nu=2, lowest=8, steps=2, warmup=1, switch_grid=128, gpus=world, transport="torch")
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        dist_bench.run(args, backend="gloo", on_gpu=False)
    with open(os.path.join(out_dir, "out%d.txt" % rank), "w") as fh:
        fh.write(buf.getvalue())


def test_multi_rank_bench_line(tmp_path):
    """bench.py's multi-GPU leg (multigridcmt_amd/dist_bench.py) rehearsed with two gloo ranks on host memory: barrier-
    bracketed timing, maximum over ranks, ONE JSON line from rank 0 with the contract's fields."""
    import json
    import torch.multiprocessing as mp
    mp.spawn(_bench_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    out0 = (tmp_path / "out0.txt").read_text()
    assert (tmp_path / "out1.txt").read_text().strip() == ""
    lines = [l for l in out0.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config"):
        assert key in out, key
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "strong" and out["config"]["strip_levels"] == 2
    assert out["value"] > 0 and abs(out["value"] - 512 * 512 * 4 * 2 / (out["ms_per_step"] * 2 * 1e-3) / 1e6) < 1e-6 * out["value"]
    # the line proves what it timed: residual history, checksum, and rank 0's single-plan run of the same right-hand side
    assert len(out["residual_reduction_per_cycle"]) == 5 and out["residual_reduction_per_cycle"][-1] < 1e-6
    par = out["parity_vs_single_plan"]
    assert par["checked"] and par["ok"], par
    assert par["max_rel_diff_checksum"] < 1e-12 and par["max_rel_diff_residual_history"] < 1e-9
    assert out["roofline"]["bound"] == "hbm" and out["roofline"]["achieved"] > 0
    assert out["config"]["exchanged_halo_rows_per_level"] == [8, 10]


def test_bench_under_torchrun(tmp_path):
    """The round driver's own launch line for N > 1 (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W`) with a rank program that swaps RCCL
    for gloo and the GPU for the emulated kernels: rendezvous through the launcher's agent store, defaults of the
    multi-GPU leg (grid / smoother overridden to CPU-sized ones), ONE JSON line from rank 0."""
    import json
    import subprocess
    import build_emu
    build_emu.build()                                 # once, before two ranks race to build it
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "bench_rehearsal.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--grid", "512", "--switch-grid", "128"]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["config"]["smoother"] == "rb" and line["config"]["grid"] == 512
    assert line["scaling"] == "strong" and line["value"] > 0
