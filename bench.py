"""bench.py — fine-grid MLUPS and V-cycles/s of the multigrid V-cycle on the 2-D Laplacian, fp64.

    python bench.py --gpus 1 --steps K --warmup W            (one MI355X)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" is one V(2,2) cycle of the hot path on a resident right-hand side (BASELINE.json
configs[2]: 2-D Laplacian 16384^2 fp64, weighted-Jacobi smoother; the cycle descends to an 8 x 8
direct solve as the reference's 2-D drivers do, 2DPotMatrixVcycle.py:95).  `value` is fine-grid MLUPS =
grid points x fine-level sweeps per cycle x cycles / wall time of the whole cycles (all levels, transfers
and the coarse solve included), so it is a lower bound on the fine smoother's own rate, which is reported
separately in `roofline` from HIP events around the fine-level sweeps alone.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
BYTES_PER_LUP = 24.0           # read v, read f, write v (fp64) per point and sweep


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--grid", type=int, default=16384)
    ap.add_argument("--smoother", default="wjacobi", choices=["wjacobi", "rb"])
    ap.add_argument("--nu", type=int, default=2, help="pre- and post-smoothing sweeps on every level")
    ap.add_argument("--lowest", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--force-sharded", action="store_true", help="run the multi-GPU driver even with one rank (testing)")
    ap.add_argument("--switch-grid", type=int, default=None, help="multi-GPU: grid below which every rank runs the whole problem")
    ap.add_argument("--config", type=int, choices=[1, 2, 3], default=None,
                    help="a BASELINE.json config by index: 1 = 4096^2 V(2,2) red-black, 2 = 16384^2 weighted Jacobi (the default, "
                         "the one `metric` is quoted on), 3 = 32768^2 V(2,2) red-black (the multi-GPU config; fits one GPU too)")
    args = ap.parse_args()
    if args.config is not None:
        args.grid, args.smoother = {1: (4096, "rb"), 2: (16384, "wjacobi"), 3: (32768, "rb")}[args.config]
    return args


def cpu_baseline(args, kind_name):
    """The oracle's C restatement (oracle/mgcmt_oracle.c) timed on this box's host cores on a bounded
    sample of the same workload: whole V(2,2) cycles on a smaller grid, scaled per point."""
    try:
        from oracle import structured
    except Exception as e:                                     # oracle not built: report, do not fail the bench
        return {"value": None, "unit": "MLUPS", "cores": 0, "kind": "port", "sample": "unavailable: %s" % e}
    return structured.time_cpu_baseline(kind_name, args.nu, args.lowest, args.cpu_seconds, grid=min(args.grid, 8192),
                                        workload_grid=args.grid)


def measured_traffic(args):
    """HBM bytes per launch of the dominant kernel from the PMC passes of scripts/gpu_pmc.sh (rocprofv3 cannot
    profile the process it runs in; the summary of that run of this same command is kept under profiles/)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as fh:
            table = json.load(fh)
        return table.get("%s_%d" % (args.smoother, args.grid))
    except Exception:
        return None


def main():
    args = parse()
    from multigridcmt_amd import _lib
    from multigridcmt_amd.operators import laplacian_operator
    from multigridcmt_amd.plan import Plan
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 or args.force_sharded:
        from multigridcmt_amd import dist_bench
        return dist_bench.run(args)

    g = args.grid
    kind = _lib.WJACOBI if args.smoother == "wjacobi" else _lib.GS_MC
    omega = 2.0 / 3.0 if args.smoother == "wjacobi" else 1.0
    op = laplacian_operator(g, "2d") * (-1.0 / np.pi ** 2)
    plan = Plan(op, args.lowest, nvec=1, device=0)
    plan.set_shifts([0.0])
    rng = np.random.RandomState(1)
    f = rng.rand(g * g)
    plan.upload(0, _lib.SLOT_F, 0, f)
    plan.fill(0, _lib.SLOT_V, 0, 0.0)
    del f

    def cycle():
        plan.vcycle(args.nu, args.nu, kind, omega=omega, k=1, nu_coarse=args.nu)

    for _ in range(args.warmup):
        cycle()
    plan.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        cycle()
    plan.sync()
    elapsed = time.perf_counter() - t0

    n = float(g) * g
    sweeps = 2 * args.nu
    value = n * sweeps * args.steps / elapsed / 1e6
    # dominant kernel: the fine-level smoother sweep, timed alone with HIP events on the same stream
    # (one launch = one fused pass of up to 2 sweeps: algorithmic bytes per launch = 24 B x points x sweeps in it)
    reps = 25                      # >= 50 fine-level sweeps at the default nu
    fuse = plan.fused_max_sweeps(0, kind)
    launches = -(-args.nu // fuse) if fuse else args.nu * (1 if kind == _lib.WJACOBI else 4)
    ms = plan.time_smoother(0, kind, args.nu, omega, reps)
    launch_s = ms * 1e-3 / (reps * launches)
    sweep_s = ms * 1e-3 / (reps * args.nu)
    achieved = n * BYTES_PER_LUP / sweep_s / 1e9
    ms1 = plan.time_smoother(0, kind, 1, omega, reps)              # one sweep per launch: the unfused comparison
    achieved1 = n * BYTES_PER_LUP / (ms1 * 1e-3 / reps) / 1e9
    probe = {name: n * bpp / (plan.bandwidth_probe(0, k_, 1024, 5) * 1e-3) / 1e9
             for k_, name, bpp in ((0, "copy", 16), (1, "triad", 24), (2, "read", 8))}
    out = {
        "metric": "fine_grid_mlups_vcycle_2d_laplacian_fp64",
        "value": value,
        "unit": "MLUPS",
        "n_gpus": 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "2D Laplacian %d^2 fp64, V(%d,%d) %s on every level, lowest_level %d, 1xMI355X" %
                   (g, args.nu, args.nu, "weighted-Jacobi (w=2/3)" if args.smoother == "wjacobi" else "red-black Gauss-Seidel",
                    args.lowest),
                   "grid": g, "smoother": args.smoother, "nu1": args.nu, "nu2": args.nu, "lowest_level": args.lowest},
        "vcycles_per_s": args.steps / elapsed,
        "smoother_mlups": n / sweep_s / 1e6,
        "roofline": {"bound": "hbm", "kernel": "fused fine-level pass (%d %s sweep(s) per launch)" % (args.nu // launches, args.smoother),
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes_per_launch": n * BYTES_PER_LUP * args.nu / launches, "avg_launch_ms": launch_s * 1e3,
                     "fusion_depth": args.nu / launches,
                     "single_sweep_per_launch": {"achieved": achieved1, "frac": achieved1 / HBM_PEAK_GBS},
                     "measured_ceilings_GBs": probe},
        "device": _lib.device_name(0),
    }
    out["roofline"]["traffic"] = measured_traffic(args)
    # untimed: what the cycles being timed do to the residual: ||f - A v|| / ||f|| after each of
    # ten cycles from a zero start
    plan.fill(0, _lib.SLOT_V, 0, 0.0)
    V, F, T = (_lib.SLOT_V, 0), (_lib.SLOT_F, 0), (_lib.SLOT_T, 0)
    f_norm = np.sqrt(plan.dot(0, F, F))
    history = []
    for _ in range(10):
        cycle()
        plan.apply(0, V, T, with_shift=True)
        plan.axpy(0, -1.0, F, T)
        history.append(float(np.sqrt(plan.dot(0, T, T)) / f_norm))
    out["residual_reduction_per_cycle"] = history
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, args.smoother)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
