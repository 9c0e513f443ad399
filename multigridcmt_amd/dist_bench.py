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


def run(args, backend="nccl", on_gpu=True):
    """backend / on_gpu exist for the CPU rehearsal of this very function (tests/test_distributed.py: gloo, host memory,
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
    def make_plan(tr):
        return ShardedPlan(op, args.lowest, rank, world, device=local, switch_grid=getattr(args, "switch_grid", None), on_gpu=on_gpu,
                           transport=tr, unique_id=uid if tr == "rccl" else None)

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
    rows = g // world
    sp.upload_local(_lib.SLOT_F, np.random.RandomState(1 + rank).rand(rows * g))   # this rank's rows of the right-hand side
    sp.fill_local(_lib.SLOT_V, 0.0)

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
    n = float(g) * g
    if rank == 0:
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
                                   "%s halo exchange" % (g, args.nu, args.nu, args.smoother, args.lowest, world, sp.switch,
                                                         "RCCL (in libmgcmt_hip.so, overlapped with the interior launches)" if transport == "rccl" else "torch.distributed"),
                       "grid": g, "smoother": args.smoother, "nu1": args.nu, "nu2": args.nu, "lowest_level": args.lowest,
                       "parallelism": "strips%d" % world, "strip_levels": sp.strip_levels, "transport": transport},
            "vcycles_per_s": args.steps / elapsed,
        }
        print(json.dumps(out), flush=True)
    sp.close()
    dist.destroy_process_group()
