"""bench.py's multi-GPU leg: one process per GPU (torch.distributed, backend nccl = RCCL), strong scaling on a
fixed grid.  Rank 0 prints the JSON line; time is the maximum over ranks between two barriers."""
import json
import os
import time

import numpy as np


def _init_group(backend, rank, world, local, on_gpu):
    """torch.distributed is the control plane (rendezvous through the launcher's env:// variables — under torchrun the
    agent's own store —, barriers, the maximum of the rank times); it also carries the RCCL communicator id of the data
    path from rank 0 to the others."""
    import torch
    import torch.distributed as dist
    if on_gpu:
        torch.cuda.set_device(local)
        try:
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        except TypeError:                                   # older signature without device_id
            dist.init_process_group(backend, rank=rank, world_size=world)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)


CHECK_CYCLES = 5


def load_rhs(sp, g):
    """This rank's rows of bench.py's multi-GPU right-hand side — the (g/2)^2 block-seeded random field of benchdata.py
    interpolated to g^2 on the device, exactly what the one-GPU leg builds (bench.py: strong_scaling_base): the rank
    generates only its own coarse rows, gets the one row above them from its neighbour and interpolates its strip."""
    from . import _lib, benchdata
    gh = g // 2
    F = (_lib.SLOT_F, 0)
    sp.plan.upload(1, _lib.SLOT_F, 0, benchdata.rhs_rows(gh, sp.row_begin // 2, sp.row_end // 2))
    sp.exchange_halo((1, _lib.SLOT_F), ring=sp.emulate is not None)
    sp.plan.prolong(0, F, F)
    sp.invalidate(_lib.SLOT_F)


def single_plan_reference(args, kind, omega, device, history, checksum):
    """Rank 0, N > 1, untimed: the same CHECK_CYCLES cycles on the whole grid as ONE plan on this GPU (the 32768^2 plan is
    43 GB of the 288 GB) — the numbers the sharded run must reproduce (red-black and Jacobi are order-independent; the
    sums differ in their reduction order only)."""
    import numpy as np
    from . import _lib, benchdata
    from .operators import laplacian_operator
    from .plan import Plan
    g = args.grid
    try:
        p = Plan(laplacian_operator(g, "2d") * (-1.0 / np.pi ** 2), args.lowest, nvec=1, device=device)
        p.set_shifts([0.0])
        V, F, T = (_lib.SLOT_V, 0), (_lib.SLOT_F, 0), (_lib.SLOT_T, 0)
        p.upload(1, _lib.SLOT_F, 0, benchdata.rhs(g // 2))
        p.prolong(0, F, F)
        p.fill(0, _lib.SLOT_V, 0, 0.0)
        f_norm = np.sqrt(p.dot(0, F, F))
        ref = []
        for _ in range(CHECK_CYCLES):
            p.vcycle(args.nu, args.nu, kind, omega=omega, k=1, nu_coarse=args.nu)
            p.apply(0, V, T, with_shift=True)
            p.axpy(0, -1.0, F, T)
            ref.append(float(np.sqrt(p.dot(0, T, T)) / f_norm))
        p.fill(0, _lib.SLOT_T, 0, 1.0)
        ref_sum = [p.dot(0, V, T), p.dot(0, V, V)]
        p.close()
    except Exception as e:                                       # e.g. no room for the whole grid beside the strip plan
        return {"checked": False, "error": str(e)}
    d_hist = max(abs(a - b) / max(abs(b), 1e-300) for a, b in zip(history, ref))
    d_sum = max(abs(a - b) / max(abs(b), 1e-300) for a, b in zip(checksum, ref_sum))
    return {"checked": True, "single_plan_residual_reduction_per_cycle": ref, "single_plan_checksum": ref_sum,
            "max_rel_diff_residual_history": d_hist, "max_rel_diff_checksum": d_sum, "ok": bool(d_hist < 1e-8 and d_sum < 1e-10)}


def time_rank_share(g, nu, lowest, smoother, erank, of, steps=10, warmup=3, device=0, switch_grid=None):
    """ONE GPU, no process group: milliseconds per cycle of rank `erank`'s share of an `of`-rank job on the g^2 grid —
    the strip plan of its rows with a one-rank RCCL communicator in self-ring mode (MGCMT_COMM_OPT_EMULATE_OF): the strip
    passes with their boundary-first launches, the RCCL send/recv groups (to itself) on the second stream, the gather of
    `of` strips' worth of coarse data, the redundant coarse sub-cycle.  What it cannot contain is the time the bytes
    spend on xGMI links.  bench.py records it next to strong_scaling_base as `ms_per_rank_share`."""
    from . import _lib
    from .distributed import ShardedPlan, rccl_unique_id
    from .operators import laplacian_operator
    kind = _lib.WJACOBI if smoother == "wjacobi" else _lib.GS_MC
    omega = 2.0 / 3.0 if smoother == "wjacobi" else 1.0
    op = laplacian_operator(g, "2d") * (-1.0 / np.pi ** 2)
    sp = ShardedPlan(op, lowest, 0, 1, device=device, switch_grid=switch_grid, transport="rccl", unique_id=rccl_unique_id(),
                     emulate=(erank, of))
    try:
        sp.set_shift(0.0)
        load_rhs(sp, g)
        sp.fill_local(_lib.SLOT_V, 0.0)
        for _ in range(warmup):
            sp.vcycle(nu, nu, kind, omega=omega, nu_coarse=nu)
        sp.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            sp.vcycle(nu, nu, kind, omega=omega, nu_coarse=nu)
        sp.sync()
        ms = (time.perf_counter() - t0) / steps * 1e3
        return {"rank": erank, "of": of, "ms_per_rank_share": ms, "strip_rows": sp.row_end - sp.row_begin, "strip_levels": sp.strip_levels,
                "switch_grid": sp.switch,
                "exchanged_halo_rows_per_level": [sp.plan.level_halo(l)[1] for l in range(sp.strip_levels)]}
    finally:
        sp.close()


def run(args, backend="nccl", on_gpu=True, cpu_baseline=None):
    """cpu_baseline: a callable returning bench.py's `cpu_baseline` record (called on rank 0, untimed).  backend / on_gpu exist for the CPU rehearsal of this very function (tests/test_distributed.py: gloo, host memory,
    emulated kernels); bench.py always calls it with the defaults."""
    import torch
    import torch.distributed as dist
    from . import _lib
    from .distributed import ShardedPlan, rccl_unique_id
    from .operators import laplacian_operator

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("MASTER_PORT", "29533")
    local = int(os.environ.get("LOCAL_RANK", rank))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    _init_group(backend, rank, world, local, on_gpu)
    transport = getattr(args, "transport", "rccl") if on_gpu else "torch"
    uid = None
    if transport == "rccl":
        box = [rccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        uid = bytes(box[0])
    g = args.grid
    kind = _lib.WJACOBI if args.smoother == "wjacobi" else _lib.GS_MC
    omega = 2.0 / 3.0 if args.smoother == "wjacobi" else 1.0
    op = laplacian_operator(g, "2d") * (-1.0 / np.pi ** 2)
    emulate = getattr(args, "emulate", None)           # (R, N): this ONE rank does rank R's share of an N-rank job (bench.py --emulate-rank)
    if emulate is not None and world != 1:
        raise SystemExit("--emulate-rank runs with --gpus 1")

    def make_plan(tr):
        return ShardedPlan(op, args.lowest, rank, world, device=local, switch_grid=getattr(args, "switch_grid", None), on_gpu=on_gpu,
                           transport=tr, unique_id=uid if tr == "rccl" else None, emulate=emulate)

    # The in-library RCCL transport has only ever run at world size 1 on the builder's one-GPU boxes.  If any rank cannot
    # bring it up, EVERY rank falls back to the torch.distributed transport (the same RCCL underneath, called through
    # callbacks from the library) — a different way to move the halo rows, not a different compute path; the line says which.
    sp, err = None, None
    try:
        sp = make_plan(transport)
    except Exception as e:                                   # noqa: BLE001 (reported below)
        err = e
    if transport == "rccl" and world > 1:
        ok = torch.tensor([0 if sp is None else 1], dtype=torch.int32, device=("cuda:%d" % local) if on_gpu else "cpu")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            if sp is not None:
                sp.close()
            if rank == 0:
                import sys
                print("bench: RCCL transport unavailable (%r): falling back to the torch.distributed transport" % (err,), file=sys.stderr)
            transport = "torch"
            sp = make_plan(transport)
    elif sp is None:
        raise err
    sp.set_shift(0.0)
    emulating = sp.emulate is not None
    load_rhs(sp, g)
    sp.fill_local(_lib.SLOT_V, 0.0)
    f_norm = float(sp.allreduce_sum([sp.plan.dot(0, (_lib.SLOT_F, 0), (_lib.SLOT_F, 0))])[0]) ** 0.5

    def cycle():
        sp.vcycle(args.nu, args.nu, kind, omega=omega, nu_coarse=args.nu)

    for _ in range(args.warmup):
        cycle()
    def device_sync():
        sp.sync()
        if on_gpu:
            torch.cuda.synchronize()

    device_sync()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        cycle()
    device_sync()
    dist.barrier()
    elapsed = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=("cuda:%d" % local) if on_gpu else "cpu")
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())
    # untimed — what the timed cycles compute: ||f - A v|| / ||f|| after each of CHECK_CYCLES cycles from a zero start and
    # (sum v, sum v^2) after the last, all-reduced over the ranks.  The one-GPU line carries the same numbers for the same
    # right-hand side (strong_scaling_base); for N > 1 rank 0 also runs the single plan here and reports the difference:
    # a run with wrong halo rows cannot print a clean line.
    sp.fill_local(_lib.SLOT_V, 0.0)
    history = []
    for _ in range(CHECK_CYCLES):
        cycle()
        history.append(sp.residual_norm() / f_norm)
    checksum = [float(x) for x in sp.checksum()]
    parity = None
    if world > 1 and not getattr(args, "no_verify", False):
        parity = single_plan_reference(args, kind, omega, local, history, checksum) if rank == 0 else None
        dist.barrier()
    # this rank's fine-level pass alone (HIP events on the stream it runs on): the strip's share of the roofline record
    n_strip = float(sp.plan.size(0))
    reps = 10
    fuse = max(sp.plan.fused_max_sweeps(0, kind), 1)
    launches = -(-args.nu // fuse)
    try:                                                         # (local to the rank: a failure here must not cost the line)
        ms = sp.plan.time_smoother(0, kind, args.nu, omega, reps)
        launch_s = ms * 1e-3 / (reps * launches)
    except Exception:                                            # noqa: BLE001
        launch_s = None
    sp.invalidate(_lib.SLOT_V)
    n = float(g) * g
    if rank == 0:
        names = {"wjacobi": "weighted-Jacobi (w=2/3)", "rb": "red-black Gauss-Seidel"}
        out = {
            "metric": "fine_grid_mlups_vcycle_2d_laplacian_fp64",
            "value": n * 2 * args.nu * args.steps / elapsed / 1e6,
            "unit": "MLUPS",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "2D Laplacian %d^2 fp64, V(%d,%d) %s, lowest_level %d, %dxMI355X row strips down to %d^2, "
                                   "%s halo exchange" % (g, args.nu, args.nu, names[args.smoother], args.lowest, world, sp.switch,
                                                         "RCCL (in libmgcmt_hip.so, overlapped with the interior launches)" if transport == "rccl" else "torch.distributed"),
                       "grid": g, "smoother": args.smoother, "nu1": args.nu, "nu2": args.nu, "lowest_level": args.lowest,
                       "parallelism": "strips%d" % world, "strip_levels": sp.strip_levels, "transport": transport,
                       "exchanged_halo_rows_per_level": [sp.plan.level_halo(l)[1] for l in range(sp.strip_levels)]},
            "vcycles_per_s": args.steps / elapsed,
            "residual_reduction_per_cycle": history,
            "checksum_after_%d_cycles" % CHECK_CYCLES: checksum,
            "roofline": {"bound": "hbm", "kernel": "fused fine-level pass on this rank's strip (%d x %d points), %d %s sweep(s) per launch"
                                                   % (sp.plan.shapes[0][0], g, min(fuse, args.nu), args.smoother),
                         "achieved": n_strip * 24.0 / launch_s / 1e9 if launch_s else None, "peak": 8000.0, "unit": "GB/s",
                         "frac": n_strip * 24.0 / launch_s / 1e9 / 8000.0 if launch_s else None, "bytes_per_launch": n_strip * 24.0,
                         "avg_launch_ms": launch_s * 1e3 if launch_s else None, "traffic": None, "rank": 0},
        }
        if parity is not None:
            out["parity_vs_single_plan"] = parity
        if emulating:
            out["emulated_rank"] = {"rank": sp.emulate[0], "of": sp.emulate[1],
                                    "note": "ONE GPU doing rank %d's share of a %d-rank job with itself as both neighbours "
                                            "(MGCMT_COMM_OPT_EMULATE_OF): strip passes, exchanges, gather, redundant coarse sub-cycle are "
                                            "timed; the halo rows hold the rank's own rows, so residuals and checksums are NOT the job's"
                                            % (sp.emulate[0], sp.emulate[1])}
            out["ms_per_rank_share"] = out["ms_per_step"]
        if cpu_baseline is not None:                               # bench.py owns that leg (the CPU checker is not importable from this package)
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    sp.close()
    dist.destroy_process_group()
    return None
