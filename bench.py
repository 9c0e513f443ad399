"""bench.py — fine-grid MLUPS and V-cycles/s of the multigrid V-cycle on the 2-D Laplacian, fp64.

    python bench.py --gpus 1 --steps K --warmup W            (one MI355X)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...                             (no launcher: starts the N ranks itself)

N = 1: one "step" is one V(2,2) cycle of the hot path on a resident right-hand side (BASELINE.json configs[2]:
2-D Laplacian 16384^2 fp64, weighted-Jacobi smoother; the cycle descends to an 8 x 8 direct solve as the
reference's 2-D drivers do, 2DPotMatrixVcycle.py:95).  `value` is fine-grid MLUPS = grid points x fine-level sweeps
per cycle x cycles / wall time of the whole cycles (all levels, transfers and the coarse solve included), so it is a
lower bound on the fine smoother's own rate, which `roofline` reports from HIP events around the fine-level passes
alone.  The same line also carries the red-black pass (`roofline_rb`, north_star's target kernel), the
reference-faithful cycle (V(4,4) below the top level, MGCMTSolver.py:320), the red-black cycle, the one-GPU time of
the multi-GPU configuration (`strong_scaling_base`), the cycle on the Mehrstellen operator, the reference's default
lexicographic Gauss-Seidel cycle at 4096^2 (`cycle_gauss_seidel_lexicographic`) and the CPU baselines.

N > 1: BASELINE.json configs[3] — 32768^2, V(2,2) red-black, row strips with RCCL halo exchange (strong scaling:
the grid is fixed as N grows; multigridcmt_amd/dist_bench.py).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
BYTES_PER_LUP = 24.0           # read v, read f, write v (fp64) per point and sweep


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--grid", type=int, default=None, help="default: 16384 on one GPU, 32768 on several")
    ap.add_argument("--smoother", default=None, choices=["wjacobi", "rb"], help="default: wjacobi on one GPU, rb on several")
    ap.add_argument("--nu", type=int, default=2, help="pre- and post-smoothing sweeps on every level")
    ap.add_argument("--lowest", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the headline cycle and its roofline record")
    ap.add_argument("--skip", default="", help="comma-separated extra legs to leave out: other, mehrstellen, lex, 1d, scaling, configs")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--force-sharded", action="store_true", help="run the multi-GPU driver even with one rank (testing)")
    ap.add_argument("--switch-grid", type=int, default=None, help="multi-GPU: grid below which every rank runs the whole problem")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "torch"],
                    help="multi-GPU halo exchange: RCCL inside libmgcmt_hip.so (default) or torch.distributed point-to-point")
    ap.add_argument("--emulate-rank", type=int, default=None, metavar="R",
                    help="with --gpus 1: this GPU does rank R's share of an --of N rank job on the multi-GPU grid (strip passes, "
                         "RCCL exchanges with itself, gather, redundant coarse sub-cycle), timed like the N-rank leg")
    ap.add_argument("--of", type=int, default=8, help="job size for --emulate-rank")
    ap.add_argument("--no-verify", action="store_true", help="multi-GPU: skip rank 0's single-plan reference run")
    ap.add_argument("--config", type=int, choices=[1, 2, 3], default=None,
                    help="a BASELINE.json config by index: 1 = 4096^2 V(2,2) red-black, 2 = 16384^2 weighted Jacobi (the default, "
                         "the one `metric` is quoted on), 3 = 32768^2 V(2,2) red-black (the multi-GPU config; fits one GPU too)")
    args = ap.parse_args(argv)
    if args.config is not None:
        args.grid, args.smoother = {1: (4096, "rb"), 2: (16384, "wjacobi"), 3: (32768, "rb")}[args.config]
    many = args.gpus > 1 or args.emulate_rank is not None
    args.emulate = None if args.emulate_rank is None else (args.emulate_rank, args.of)
    if args.emulate is not None:
        args.force_sharded = True
    if args.grid is None:
        args.grid = 32768 if many else 16384
    if args.smoother is None:
        args.smoother = "rb" if many else "wjacobi"
    return args


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (this process has not
    touched the GPU), relay rank 0's JSON line, fail if any rank fails."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0 = procs[0].communicate()[0]
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out0)
    sys.stdout.flush()
    if any(codes):
        raise SystemExit("rank exit codes: %s" % codes)


def cpu_baseline(args, kind_name):
    """The oracle's C restatement (oracle/mgcmt_oracle.c) timed on this box's host cores on the SAME workload (whole
    V(nu,nu) cycles on the full grid while the budget lasts, at least one after the warm-up cycle)."""
    try:
        from oracle import structured
    except Exception as e:                                     # oracle not built: report, do not fail the bench
        return {"value": None, "unit": "MLUPS", "cores": 0, "kind": "port", "sample": "unavailable: %s" % e}
    return structured.time_cpu_baseline(kind_name, args.nu, args.lowest, args.cpu_seconds, grid=args.grid,
                                        workload_grid=args.grid)


def cpu_reference_equivalent(grid=128):
    """the reference's own formulation on the host — generic scipy.sparse operators, the smoother's
    iteration matrix materialised by a sparse solve with a sparse right-hand side (MGCMTSolver.py:193-206), Galerkin
    products R*A*P on every level of every call (:318) — at the largest size the reference's author could run (128^2,
    BASELINE.md §1/§2: 23.6 s per V(4,4) cycle measured with the reference itself).  Restated in oracle/sparse_ref.py;
    the reference's files do not exist on this box.  Next to it the same cycle with O(N) sweeps (the oracle the
    parity tests use)."""
    try:
        from oracle.sparse_ref import RefSolver, RefStencilMaker
    except Exception as e:
        return {"value": None, "sample": "unavailable: %s" % e}
    ref, rsm = RefSolver(), RefStencilMaker()
    A = (-1 / np.pi ** 2) * rsm.laplacian(grid, dimension="2d")
    f = np.random.RandomState(1).rand(grid * grid)
    t0 = time.perf_counter()
    ref.vcycle(np.zeros(grid * grid), f, A, rsm, nu1=4, nu2=4, dimension="2d", lowest_level=8, smoother=ref.wjacobi_iteration_matrix)
    t_ref = time.perf_counter() - t0
    t0 = time.perf_counter()
    ref.vcycle(np.zeros(grid * grid), f, A, rsm, nu1=4, nu2=4, dimension="2d", lowest_level=8)
    t_fast = time.perf_counter() - t0
    lups = float(grid * grid) * 8
    return {"value": lups / t_ref / 1e6, "unit": "MLUPS", "cores": 1, "kind": "port",
            "seconds_per_vcycle": t_ref,
            "sample": "one V(4,4) weighted-Jacobi cycle on %d^2 (lowest_level 8) in the reference's formulation: scipy.sparse "
                      "operators, iteration matrix by spsolve with a sparse right-hand side, R*A*P per level "
                      "(oracle/sparse_ref.py wjacobi_iteration_matrix); single-threaded SciPy" % grid,
            "same_cycle_O_N_sweeps": {"value": lups / t_fast / 1e6, "seconds_per_vcycle": t_fast,
                                      "sample": "oracle/sparse_ref.py with one sparse mat-vec per sweep"}}


def traffic_record(args, smoother, nsweep):
    """HBM bytes per launch of the fused fine-level pass from the PMC passes of scripts/gpu_pmc.sh (rocprofv3 cannot
    profile the process it runs in: the figure comes from a profiled run of this same command, kept under profiles/
    together with where it came from)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as fh:
            table = json.load(fh)
        key = "%s_%d" % (smoother, args.grid)   # e.g. wjacobi_16384 (stand-alone pass), wjacobi_up_16384 (the cycle's up pass)
        val = table.get(key)
        if val is None:
            return None, None
        src = dict(table.get("_source", {}))
        src.setdefault("file", "profiles/pmc_traffic.json")
        src["key"] = key
        src["measured_in_this_run"] = False
        return val, src
    except Exception:
        return None, None


def pass_roofline(plan, args, kind, smoother, omega, n, reps=25):
    """HIP-event timing (on the stream the kernels are launched on, mgcmt_time_smoother) of the fused fine-level
    pass.  `achieved` / `frac`: the bytes ONE pass over the level has to move (read v, read f, write v' = 24 B per
    point — what the HBM actually has to deliver, whatever the number of sweeps fused into the pass) per second of
    launch time; `achieved_algorithmic_24B_per_update` / `frac_algorithmic_24B`: SURVEY §8(d)'s accounting, 24 B per
    lattice-site UPDATE, which counts the pass's bytes once per fused sweep and so exceeds 1 at fusion depth 2."""
    from multigridcmt_amd import _lib
    fuse = plan.fused_max_sweeps(0, kind)
    nu = args.nu
    launches = -(-nu // fuse) if fuse else nu * (1 if kind == _lib.WJACOBI else 4)
    ms = plan.time_smoother(0, kind, nu, omega, reps)
    launch_s = ms * 1e-3 / (reps * launches)
    depth = nu / launches
    phys = n * BYTES_PER_LUP / launch_s / 1e9
    algo = n * BYTES_PER_LUP * depth / launch_s / 1e9
    ms1 = plan.time_smoother(0, kind, 1, omega, reps)              # one sweep per launch
    one = n * BYTES_PER_LUP / (ms1 * 1e-3 / reps) / 1e9
    traffic, source = traffic_record(args, smoother, int(depth))
    rec = {"bound": "hbm",
           "kernel": "k_fused<Op5,%d,%d,0>: fused fine-level pass, %d %s sweep(s) per launch" % (0 if kind == _lib.WJACOBI else 1, int(depth), int(depth), smoother),
           "achieved": phys, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": phys / HBM_PEAK_GBS,
           "bytes_per_launch": n * BYTES_PER_LUP, "avg_launch_ms": launch_s * 1e3,
           "achieved_algorithmic_24B_per_update": algo, "frac_algorithmic_24B": algo / HBM_PEAK_GBS,
           "algorithmic_bytes_per_launch": n * BYTES_PER_LUP * depth, "fusion_depth": depth,
           "single_sweep_per_launch": {"achieved": one, "frac": one / HBM_PEAK_GBS, "avg_launch_ms": ms1 / reps},
           "smoother_mlups": n * depth / launch_s / 1e6,
           "traffic": traffic, "traffic_source": source}
    return rec


def cycle_compulsory_bytes(plan, nu, kind):
    """Bytes one V(nu,nu) cycle HAS to move, level by level, with the plan's own traffic-saving modes:
    a down pass reads V (not below the top level: the coarse iterate starts at zero) and F, writes V' (not on levels of
    >= 2^22 points: recompute instead of store) and the coarse right-hand side (2 B per fine point); an up pass reads V
    (unless it is still the untouched zero), F and the coarse correction and writes V'.  Only levels on the fused passes
    count (the LDS-resident tail below moves a few KB)."""
    total = 0.0
    for l in range(plan.num_levels - 1):
        r, c, _ = plan.shapes[l]
        pts = float(r) * c
        if plan.fused_max_sweeps(l, kind) == 0:
            break
        recompute = pts >= (1 << 22) and nu <= 2
        top = l == 0
        down = (8 if top else 0) + 8 + (0 if recompute else 8) + 2
        up = (8 if (top or not recompute) else 0) + 8 + 2 + 8
        total += pts * (down + up)
    return total


def cycle_roofline(plan, args, kind, smoother, omega, n, ms_per_step, reps=20):
    """The roofline record of the timed region: the fine level's down pass (pre-smoothing + residual + restriction, V' not
    stored: 18 B per point compulsory) and up pass (pre-smoothing recomputed + correction + post-smoothing: 26 B) timed
    with HIP events on the stream they run on (mgcmt_time_fused_pass), the slower one as the record's kernel, and the
    whole cycle's compulsory bytes over its measured time."""
    from multigridcmt_amd import _lib
    nu = args.nu
    multicolour = kind != _lib.WJACOBI
    passes = []
    try:
        npre = min(nu, plan.fused_max_recompute(0, kind, min(nu, 2)))
        down_mode = 2 | (8 if npre == nu else 0)
        up_mode = 1 | ((npre if npre == nu else 0) << 4)
        kid = 1 if multicolour else 0
        for name, mode, bpp, flags in (("down", down_mode, 18.0 if npre == nu else 26.0, down_mode & 15),
                                       ("up", up_mode, 26.0, (up_mode & 15) | ((up_mode >> 4) << 4))):
            ms = plan.time_fused_pass(0, kind, min(nu, 2), omega, mode, reps)
            passes.append({"pass": name, "kernel": "k_fused<Op5,%d,%d,%d>" % (kid, min(nu, 2), flags), "avg_launch_ms": ms,
                           "compulsory_bytes_per_launch": n * bpp, "achieved": n * bpp / (ms * 1e-3) / 1e9,
                           "frac": n * bpp / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS})
        plan.fill(0, _lib.SLOT_V, 0, 0.0)               # (the timed passes left V arbitrary)
    except Exception as e:                              # a plan whose fine level is not on the fused passes
        return {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None, "error": str(e)}
    dom = max(passes, key=lambda r: r["avg_launch_ms"])
    traffic, source = traffic_record(args, smoother + "_" + dom["pass"], 0)
    cyc = cycle_compulsory_bytes(plan, nu, kind)
    return {"bound": "hbm", "kernel": "%s: the fine level's %s pass of the timed cycle" % (dom["kernel"], dom["pass"]),
            "achieved": dom["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["frac"],
            "bytes_per_launch": dom["compulsory_bytes_per_launch"], "avg_launch_ms": dom["avg_launch_ms"],
            "traffic": traffic, "traffic_source": source, "passes": passes,
            "cycle_compulsory_bytes": cyc, "cycle_compulsory_frac": cyc / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS}


def time_cycles(plan, steps, warmup, cycle):
    for _ in range(warmup):
        cycle()
    plan.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        cycle()
    plan.sync()
    return time.perf_counter() - t0


def main(argv=None):
    args = parse(argv)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 or args.force_sharded:
        from multigridcmt_amd import dist_bench
        cpu = None
        if not args.no_cpu_baseline:
            def cpu():
                try:
                    from oracle import structured
                    return structured.time_cpu_baseline(args.smoother, args.nu, args.lowest, args.cpu_seconds, grid=min(args.grid, 8192),
                                                        workload_grid=args.grid)
                except Exception as e:                         # oracle not built on this box: report, do not fail the bench
                    return {"value": None, "unit": "MLUPS", "cores": 0, "kind": "port", "sample": "unavailable: %s" % e}
        return dist_bench.run(args, cpu_baseline=cpu)
    from multigridcmt_amd import _lib
    from multigridcmt_amd.operators import laplacian_operator
    from multigridcmt_amd.plan import Plan

    g = args.grid
    skipped = set(x for x in args.skip.split(",") if x)

    def leg(name):
        return not args.no_extras and name not in skipped

    kinds = {"wjacobi": (_lib.WJACOBI, 2.0 / 3.0), "rb": (_lib.GS_MC, 1.0)}
    kind, omega = kinds[args.smoother]
    op = laplacian_operator(g, "2d") * (-1.0 / np.pi ** 2)
    plan = Plan(op, args.lowest, nvec=1, device=0)
    plan.set_shifts([0.0])
    from multigridcmt_amd import benchdata
    f = benchdata.rhs(g)                       # block-seeded uniform(0,1) field: a multi-GPU rank can produce its own rows of it
    plan.upload(0, _lib.SLOT_F, 0, f)
    plan.fill(0, _lib.SLOT_V, 0, 0.0)

    def cycle(kind_=kind, omega_=omega, nu_coarse=None):
        plan.vcycle(args.nu, args.nu, kind_, omega=omega_, k=1, nu_coarse=args.nu if nu_coarse is None else nu_coarse)

    elapsed = time_cycles(plan, args.steps, args.warmup, cycle)

    n = float(g) * g
    sweeps = 2 * args.nu
    value = n * sweeps * args.steps / elapsed / 1e6
    names = {"wjacobi": "weighted-Jacobi (w=2/3)", "rb": "red-black Gauss-Seidel"}
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
                   (g, args.nu, args.nu, names[args.smoother], args.lowest),
                   "grid": g, "smoother": args.smoother, "nu1": args.nu, "nu2": args.nu, "lowest_level": args.lowest},
        "vcycles_per_s": args.steps / elapsed,
        "device": _lib.device_name(0),
    }
    # dominant kernels of the TIMED region: the fine level's two transfer-fused passes (the stand-alone smoother pass,
    # which the cycle does not run, is `roofline_smoother`)
    out["roofline"] = cycle_roofline(plan, args, kind, args.smoother, omega, n, out["ms_per_step"])
    out["roofline_smoother"] = pass_roofline(plan, args, kind, args.smoother, omega, n)
    out["smoother_mlups"] = out["roofline_smoother"]["smoother_mlups"]
    out["roofline"]["measured_ceilings_GBs"] = {
        name: n * bpp / (plan.bandwidth_probe(0, k_, 1024, 5) * 1e-3) / 1e9
        for k_, name, bpp in ((0, "copy", 16), (1, "triad", 24), (2, "read", 8))}
    if leg('other'):
        other = "rb" if args.smoother == "wjacobi" else "wjacobi"
        okind, oomega = kinds[other]
        # north_star's target kernel (the red-black fine-grid sweep) when the headline smoother is weighted Jacobi,
        # and the other way round
        out["roofline_" + other] = pass_roofline(plan, args, okind, other, oomega, n)
        # the reference-faithful cycle: V(nu,nu) on the finest level, V(4,4) below it (MGCMTSolver.py:320 does not
        # forward nu1/nu2; BASELINE.md §4's 208 B/fine point row)
        plan.fill(0, _lib.SLOT_V, 0, 0.0)
        steps = max(3, min(args.steps, 10))
        t = time_cycles(plan, steps, 2, lambda: cycle(nu_coarse=4))
        out["vcycles_per_s_reference_faithful"] = steps / t
        out["reference_faithful_cycle"] = {"workload": "V(%d,%d) on the finest level, V(4,4) below (MGCMTSolver.py:320)" % (args.nu, args.nu),
                                           "ms_per_step": t / steps * 1e3, "value": n * sweeps * steps / t / 1e6, "unit": "MLUPS"}
        plan.fill(0, _lib.SLOT_V, 0, 0.0)
        t = time_cycles(plan, steps, 2, lambda: cycle(okind, oomega))
        out["cycle_" + other] = {"workload": "V(%d,%d) %s on every level" % (args.nu, args.nu, names[other]),
                                 "ms_per_step": t / steps * 1e3, "vcycles_per_s": steps / t, "value": n * sweeps * steps / t / 1e6, "unit": "MLUPS"}
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
    plan.close()
    if leg('mehrstellen'):
        # SURVEY par. 8(f)3: the same cycle on the fourth-order compact 9-point (Mehrstellen) fine-grid operator
        try:
            from multigridcmt_amd.operators import mehrstellen_operator
            m9 = Plan(mehrstellen_operator(g) * (-1.0 / np.pi ** 2), args.lowest, nvec=1, device=0)
            m9.set_shifts([0.0])
            m9.upload(0, _lib.SLOT_F, 0, f)
            steps = 5
            rec = {"workload": "Mehrstellen 9-point Laplacian %d^2 fp64, V(%d,%d), 1xMI355X" % (g, args.nu, args.nu)}
            for name_, (k_, om_) in (("wjacobi", kinds["wjacobi"]), ("four_colour", kinds["rb"])):
                m9.fill(0, _lib.SLOT_V, 0, 0.0)
                t = time_cycles(m9, steps, 2, lambda: m9.vcycle(args.nu, args.nu, k_, omega=om_, k=1, nu_coarse=args.nu))
                rec[name_] = {"ms_per_step": t / steps * 1e3, "vcycles_per_s": steps / t, "value": n * sweeps * steps / t / 1e6, "unit": "MLUPS"}
            out["cycle_mehrstellen"] = rec
            m9.close()
        except Exception as e:
            out["cycle_mehrstellen"] = {"value": None, "error": str(e)}
    if leg('lex'):
        # the reference's DEFAULT smoother — lexicographic Gauss-Seidel (MGCMTSolver.py:210-227; `smoother=None` ->
        # `self.gseidel`, :291) — at BASELINE config 2's grid: V(2,2), and V(2,2) on top / V(4,4) below (:320), as a
        # pipeline of waves with the sweeps of a smoothing step chained in one launch, against the one-workgroup kernel
        try:
            gl = min(g, 4096)
            lx = Plan(laplacian_operator(gl, "2d") * (-1.0 / np.pi ** 2), args.lowest, nvec=1, device=0)
            lx.set_shifts([0.0])
            lx.upload(0, _lib.SLOT_F, 0, np.random.RandomState(2).rand(gl * gl))
            rec = {"workload": "2D Laplacian %d^2 fp64, lexicographic Gauss-Seidel (the reference's default smoother), 1xMI355X" % gl}
            for name_, nuc in (("V22", 2), ("V22_top_V44_below", 4)):
                lx.fill(0, _lib.SLOT_V, 0, 0.0)
                t = time_cycles(lx, 5, 3, lambda: lx.vcycle(2, 2, _lib.GS_LEX, omega=1.0, k=1, nu_coarse=nuc))
                rec[name_ + "_ms_per_step"] = t / 5 * 1e3
            lx.set_option(_lib.OPT_LEX_WAVE, 0)
            lx.fill(0, _lib.SLOT_V, 0, 0.0)
            t = time_cycles(lx, 2, 1, lambda: lx.vcycle(2, 2, _lib.GS_LEX, omega=1.0, k=1, nu_coarse=2))
            rec["V22_one_workgroup_ms_per_step"] = t / 2 * 1e3
            rec["speedup_V22"] = rec["V22_one_workgroup_ms_per_step"] / rec["V22_ms_per_step"]
            out["cycle_gauss_seidel_lexicographic"] = rec
            lx.close()
        except Exception as e:
            out["cycle_gauss_seidel_lexicographic"] = {"value": None, "error": str(e)}
    if leg('1d'):
        # 1-D cycles (the reference's own problems are 1-D: 1DPotMatrixVcycle.py:68-75, RQMin.py, the UnitTests): the fused
        # 1-D passes (csrc/kernels_fused1d.hip) at n = 2^24, V(2,2) and the reference's own V(4,4)
        try:
            n1 = 1 << 24
            p1 = Plan(laplacian_operator(n1, "1d") * (-1.0 / np.pi ** 2), args.lowest, nvec=1, device=0)
            p1.set_shifts([0.0])
            p1.upload(0, _lib.SLOT_F, 0, f[:n1])
            rec = {"workload": "1D Laplacian n = 2^24 fp64, lowest_level %d, 1xMI355X" % args.lowest}
            for name_, (k_, om_) in (("wjacobi", kinds["wjacobi"]), ("redblack", kinds["rb"])):
                for nu_ in (2, 4):
                    p1.fill(0, _lib.SLOT_V, 0, 0.0)
                    t = time_cycles(p1, 20, 3, lambda: p1.vcycle(nu_, nu_, k_, omega=om_, k=1, nu_coarse=nu_))
                    rec["V%d%d_%s_ms_per_step" % (nu_, nu_, name_)] = t / 20 * 1e3
            # compulsory bytes of the V(2,2) cycle: 18 + 26 B per point on levels of >= 2^22 points, 26 + 26 below
            cyc1 = sum((n1 >> l) * ((18.0 + 26.0) if (n1 >> l) >= (1 << 22) else 52.0) for l in range(p1.num_levels - 1) if (n1 >> l) >= 256)
            rec["V22_compulsory_bytes"] = cyc1
            rec["V22_wjacobi_compulsory_frac"] = cyc1 / (rec["V22_wjacobi_ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS
            p1.fill(0, _lib.SLOT_V, 0, 0.0)
            p1.set_option(_lib.OPT_FUSED, 0)
            t = time_cycles(p1, 5, 2, lambda: p1.vcycle(2, 2, _lib.WJACOBI, omega=2.0 / 3.0, k=1, nu_coarse=2))
            rec["V22_wjacobi_one_launch_per_operation_ms_per_step"] = t / 5 * 1e3
            out["cycle_1d"] = rec
            p1.close()
        except Exception as e:
            out["cycle_1d"] = {"value": None, "error": str(e)}
    if leg('scaling') and g == 16384:
        # BASELINE config 4's workload (32768^2, V(2,2) red-black) on this ONE GPU: the base of the strong-scaling
        # curve `bench.py --gpus N` continues.  Right-hand side: the 16384^2 random field interpolated on the device.
        try:
            big = Plan(laplacian_operator(2 * g, "2d") * (-1.0 / np.pi ** 2), args.lowest, nvec=1, device=0)
            big.set_shifts([0.0])
            big.upload(1, _lib.SLOT_F, 0, f)
            big.prolong(0, (_lib.SLOT_F, 0), (_lib.SLOT_F, 0))
            big.fill(0, _lib.SLOT_V, 0, 0.0)
            steps = 5
            t = time_cycles(big, steps, 2, lambda: big.vcycle(args.nu, args.nu, _lib.GS_MC, omega=1.0, k=1, nu_coarse=args.nu))
            out["strong_scaling_base"] = {"workload": "2D Laplacian %d^2 fp64, V(%d,%d) red-black Gauss-Seidel, 1xMI355X (what --gpus N shards)" % (2 * g, args.nu, args.nu),
                                          "ms_per_step": t / steps * 1e3, "vcycles_per_s": steps / t,
                                          "value": 4 * n * sweeps * steps / t / 1e6, "unit": "MLUPS"}
            # what `bench.py --gpus N` must reproduce (it prints the same two records for the same right-hand side)
            from multigridcmt_amd.dist_bench import CHECK_CYCLES
            Vb, Fb, Tb = (_lib.SLOT_V, 0), (_lib.SLOT_F, 0), (_lib.SLOT_T, 0)
            big.fill(0, _lib.SLOT_V, 0, 0.0)
            fb_norm = np.sqrt(big.dot(0, Fb, Fb))
            hist = []
            for _ in range(CHECK_CYCLES):
                big.vcycle(args.nu, args.nu, _lib.GS_MC, omega=1.0, k=1, nu_coarse=args.nu)
                big.apply(0, Vb, Tb, with_shift=True)
                big.axpy(0, -1.0, Fb, Tb)
                hist.append(float(np.sqrt(big.dot(0, Tb, Tb)) / fb_norm))
            big.fill(0, _lib.SLOT_T, 0, 1.0)
            out["strong_scaling_base"]["residual_reduction_per_cycle"] = hist
            out["strong_scaling_base"]["checksum_after_%d_cycles" % CHECK_CYCLES] = [big.dot(0, Vb, Tb), big.dot(0, Vb, Vb)]
            big.close()
            # ONE rank's share of the 8-rank job, on this GPU (an edge and an interior rank): strip passes, RCCL exchanges
            # with itself, gather, redundant coarse sub-cycle — everything but the bytes' time on the xGMI links
            try:
                from multigridcmt_amd.dist_bench import time_rank_share
                shares = [time_rank_share(2 * g, args.nu, args.lowest, "rb", r, 8) for r in (0, 3)]
                out["strong_scaling_base"]["rank_share_of_8"] = shares
                out["strong_scaling_base"]["ms_per_rank_share"] = max(x["ms_per_rank_share"] for x in shares)
                out["strong_scaling_base"]["speedup_bound_from_rank_share"] = out["strong_scaling_base"]["ms_per_step"] / out["strong_scaling_base"]["ms_per_rank_share"]
            except Exception as e:                              # RCCL not loadable on this box
                out["strong_scaling_base"]["rank_share_of_8"] = {"value": None, "error": str(e)}
        except Exception as e:                                  # e.g. a smaller-memory device
            out["strong_scaling_base"] = {"value": None, "error": str(e)}
    if leg('configs') and g == 16384:
        # BASELINE configs 2 and 5 in the same line (their own bench commands: `--config 1`, scripts/bench_config5.py)
        try:
            g2 = 4096
            c2 = Plan(laplacian_operator(g2, "2d") * (-1.0 / np.pi ** 2), args.lowest, nvec=1, device=0)
            c2.set_shifts([0.0])
            c2.upload(0, _lib.SLOT_F, 0, f[:g2 * g2])
            c2.fill(0, _lib.SLOT_V, 0, 0.0)
            t = time_cycles(c2, 40, 5, lambda: c2.vcycle(2, 2, _lib.GS_MC, omega=1.0, k=1, nu_coarse=2))
            out["config2_4096_redblack"] = {"workload": "2D Laplacian 4096^2 fp64, V(2,2) red-black Gauss-Seidel, 1xMI355X",
                                            "ms_per_step": t / 40 * 1e3, "vcycles_per_s": 40 / t,
                                            "value": float(g2) * g2 * 4 * 40 / t / 1e6, "unit": "MLUPS"}
            c2.close()
        except Exception as e:
            out["config2_4096_redblack"] = {"value": None, "error": str(e)}
        try:
            from multigridcmt_amd import drivers
            from multigridcmt_amd.operators import potential_well_operator
            g5 = 8192
            rec = {"workload": "2D square-well Hamiltonian %d^2 fp64 (PotWellSolver.py:150-153 carried to 2-D): V(2,2) cycle of H; "
                               "Rayleigh-quotient minimisation with that cycle as the preconditioner, 1xMI355X" % g5}
            w5 = Plan(potential_well_operator(g5, 50.0, (g5 // 4, 3 * g5 // 4)), args.lowest, nvec=1, device=0)
            w5.set_shifts([0.0])
            w5.upload(0, _lib.SLOT_F, 0, f[:g5 * g5])
            for name_, (k_, om_) in (("wjacobi", kinds["wjacobi"]), ("redblack", kinds["rb"])):
                w5.fill(0, _lib.SLOT_V, 0, 0.0)
                t = time_cycles(w5, 10, 3, lambda: w5.vcycle(2, 2, k_, omega=om_, k=1, nu_coarse=2))
                rec["vcycle_%s_ms_per_step" % name_] = t / 10 * 1e3
            w5.close()
            drivers.potential_well_eigensolve(g5, cycles=2, method="vcycle")            # plan, graph capture
            stats, hist = {}, []
            drivers.potential_well_eigensolve(g5, cycles=10, method="vcycle", stats=stats, history=hist)
            rec["eigen_iteration_ms"] = stats["loop_seconds"] / 10 * 1e3
            rec["rayleigh_quotient_history"] = hist
            out["config5_8192_square_well"] = rec
            from multigridcmt_amd.plan import release_plans
            release_plans()
        except Exception as e:
            out["config5_8192_square_well"] = {"value": None, "error": str(e)}
    del f
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, args.smoother)
        if not args.no_extras:
            out["cpu_reference_equivalent"] = cpu_reference_equivalent(128)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
