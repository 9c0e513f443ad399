"""Turn one run of scripts/gpu_r03_profiles.sh (gpurun_out/*_<tag>*) into the tracked summaries under profiles/:

  pmc_traffic.json                  HBM bytes per launch of the fused fine-level pass (what bench.py reports as `traffic`),
                                    with `_source` = where the numbers came from
  <prefix>_pmc_traffic_detail.json  FETCH_SIZE (x2 on gfx950, MI355X_MICROARCH.md "HBM") / WRITE_SIZE per kernel
  <prefix>_kernel_table.md          per kernel of the 16384^2 cycles: launches, average time, VGPRs, waves/SIMD the
                                    registers allow, VALU-active share of the wave cycles, HBM bytes and the rate they imply
  <prefix>_{wjacobi,rb,cfg2,cfg5}_kernel_stats.csv   rocprofv3 --stats tables
usage: collect_profiles_r02.py <tag> <out-prefix> [commit]"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, prefix = sys.argv[1], sys.argv[2]
commit = sys.argv[3] if len(sys.argv) > 3 else subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
out = os.path.join(ROOT, "profiles")
g = os.path.join(ROOT, "gpurun_out")
n = 16384


def latest(pattern):
    files = sorted(glob.glob(os.path.join(g, pattern), recursive=True), key=os.path.getmtime)
    return files[-1] if files else None


def short(name):
    name = name.replace("void ", "").replace("mgcmt::fused::", "").replace("mgcmt::(anonymous namespace)::", "").replace("mgcmt::", "")
    return name.split("(")[0]


# The kernel-trace CSV's VGPR_Count is HALF the allocation on this stack: for k_fused<Op9c, 0, 2, 14> the compiler's own
# .vgpr_count / "; Occupancy" say 117 registers, 4 waves per SIMD (hipcc -S --cuda-device-only of kernels_fused_op9c.hip),
# the trace says 60; the same factor holds for every kernel compared (Op5 plain pass 104 vs 52, red-black up-leg pass
# 216 vs 108).  The table carries the allocation (2 x the traced value) and the waves per SIMD that follow from it.
TRACE_VGPR_FACTOR = 2


def vgpr_alloc(traced):
    return -(-int(traced) * TRACE_VGPR_FACTOR // 8) * 8


def waves_per_simd(traced):
    return max(1, min(8, 512 // max(vgpr_alloc(traced), 1)))


def clusters(values, ratio=0.62):
    """Group dispatches of one kernel by size: the same instantiation runs on several levels, each about four times smaller
    than the one above (two to three times shorter).  Returns for every value the index of its cluster (0 = largest)."""
    order = sorted(set(values), reverse=True)
    bounds, top = [], None
    for x in order:
        if top is None or x < ratio * top:
            bounds.append(x)
            top = x
    def index(x):
        k = 0
        for b, bound in enumerate(bounds):
            if x <= bound * 1.0000001:
                k = b
        return k
    return [index(x) for x in values]


table_rows, detail, traffic = [], {}, {}
for sm in ("wjacobi", "rb"):
    per = {}
    trace = latest("prof_%s_%s/**/*kernel_trace.csv" % (tag, sm))
    if trace:
        rows = [r for r in csv.DictReader(open(trace))]
        by_kernel = {}
        for r in rows:
            by_kernel.setdefault(r["Kernel_Name"], []).append(r)
        for kname, rs in by_kernel.items():
            durs = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in rs]
            lev = clusters(durs) if "k_fused" in kname else [0] * len(durs)
            for r, d, l in zip(rs, durs, lev):
                k = per.setdefault((kname, l), {"calls": 0, "ns": 0.0, "vgpr": r.get("VGPR_Count"), "sgpr": r.get("SGPR_Count"), "lds": r.get("LDS_Block_Size")})
                k["calls"] += 1
                k["ns"] += d
    for counter_dir, counters in (("FETCH_SIZE", ["FETCH_SIZE"]), ("WRITE_SIZE", ["WRITE_SIZE"]),
                                  ("SQ", ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_WAIT_ANY",
                                          "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"])):
        f = latest("pmc_%s_%s_%s/**/*counter_collection.csv" % (tag, sm, counter_dir))
        if not f:
            continue
        rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] in counters]
        # level of a dispatch: from the counter that scales with the level's size (bytes moved / wave cycles)
        key_counter = counters[0] if counter_dir != "SQ" else "SQ_WAVE_CYCLES"
        size_of = {}
        for r in rows:
            if r["Counter_Name"] == key_counter:
                size_of.setdefault(r["Kernel_Name"], {})[r["Dispatch_Id"]] = float(r["Counter_Value"])
        level_of = {}
        for kname, d in size_of.items():
            ids, vals = list(d.keys()), list(d.values())
            lev = clusters(vals) if "k_fused" in kname else [0] * len(vals)
            for i, l in zip(ids, lev):
                level_of[(kname, i)] = l
        acc = {}
        for r in rows:
            l = level_of.get((r["Kernel_Name"], r["Dispatch_Id"]), 0)
            a = acc.setdefault((r["Kernel_Name"], l, r["Counter_Name"]), [0.0, 0])
            a[0] += float(r["Counter_Value"])
            a[1] += 1
        for (kname, l, cname), (tot, cnt) in acc.items():
            per.setdefault((kname, l), {"calls": 0, "ns": 0.0, "vgpr": None, "sgpr": None, "lds": None})[cname] = tot / cnt
    for (kname, lev), k in sorted(per.items(), key=lambda kv: -kv[1]["ns"]):
        if not k["calls"] or "k_probe" in kname or "rocclr" in kname:
            continue
        avg_us = k["ns"] / k["calls"] / 1e3
        fetch = k.get("FETCH_SIZE")
        write = k.get("WRITE_SIZE")
        hbm = (fetch * 2048 + write * 1024) if fetch is not None and write is not None else None
        valu = (100.0 * k["SQ_ACTIVE_INST_VALU"] / k["SQ_WAVE_CYCLES"]) if k.get("SQ_WAVE_CYCLES") else None
        wait = (100.0 * k["SQ_WAIT_ANY"] / k["SQ_WAVE_CYCLES"]) if k.get("SQ_WAVE_CYCLES") else None
        label = short(kname) + (" [size class %d]" % lev if "k_fused" in kname else "")
        table_rows.append((sm, label, k["calls"], avg_us, vgpr_alloc(k["vgpr"]) if k["vgpr"] else None, waves_per_simd(k["vgpr"]) if k["vgpr"] else None, valu, wait,
                           hbm, (hbm / (avg_us * 1e-6) / 1e12) if hbm else None))
        if hbm is not None:
            detail["%s_%d %s" % (sm, n, label)] = {"avg_us": avg_us, "FETCH_SIZE_KB_avg": fetch, "WRITE_SIZE_KB_avg": write,
                                                   "fetch_bytes_corrected_x2": fetch * 2048, "write_bytes": write * 1024, "hbm_bytes": hbm}
        kid = 0 if sm == "wjacobi" else 1
        # the stand-alone smoother pass and the two transfer-fused passes of the timed cycle (bench.py: roofline_smoother / roofline)
        for flags, key in ((0, "%s_%d" % (sm, n)), (10, "%s_down_%d" % (sm, n)), (33, "%s_up_%d" % (sm, n))):
            if short(kname) == "k_fused<Op5, %d, 2, %d>" % (kid, flags) and lev == 0 and hbm is not None:
                traffic[key] = hbm
    stats = latest("prof_%s_%s/**/*kernel_stats.csv" % (tag, sm))
    if stats:
        shutil.copy(stats, os.path.join(out, "%s_%s_%d_kernel_stats.csv" % (prefix, sm, n)))
    log = os.path.join(g, "prof_%s_%s.log" % (tag, sm))
    if os.path.exists(log):
        lines = [l for l in open(log) if l.startswith('{"metric"')]
        if lines:
            open(os.path.join(out, "%s_bench_under_rocprof_%s.json" % (prefix, sm)), "w").write(lines[-1])
for cfg in ("cfg2", "cfg5", "1d", "rqmg", "share"):
    stats = latest("prof_%s_%s/**/*kernel_stats.csv" % (tag, cfg))
    if stats:
        shutil.copy(stats, os.path.join(out, "%s_%s_kernel_stats.csv" % (prefix, cfg)))
    log = os.path.join(g, "prof_%s_%s.log" % (tag, cfg))
    if os.path.exists(log):
        lines = [l for l in open(log) if l.startswith("{")]
        if lines:
            open(os.path.join(out, "%s_%s_bench_under_rocprof.json" % (prefix, cfg)), "w").write("".join(lines))
if traffic:
    traffic["_source"] = {"file": "profiles/pmc_traffic.json", "commit": commit, "run_tag": tag,
                          "kernel": "k_fused<Op5,{0|1},2,F>: F = 0 the stand-alone fine-level pass (<sm>_16384), 10 the cycle's down pass (<sm>_down_16384), 33 its up pass (<sm>_up_16384)",
                          "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --steps 2 --warmup 1 --smoother <sm> (scripts/gpu_r03_profiles.sh)",
                          "correction": "FETCH_SIZE x 2 (gfx950 tallies 128-byte requests of 16-byte streams at 64 B, MI355X_MICROARCH.md HBM), counters in KB"}
    json.dump(traffic, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
    json.dump(detail, open(os.path.join(out, "%s_pmc_traffic_detail.json" % prefix), "w"), indent=1)
with open(os.path.join(out, "%s_kernel_table.md" % prefix), "w") as fh:
    fh.write("# Per-kernel table of the 16384^2 V(2,2) cycles (run tag %s, commit %s)\n\n" % (tag, commit))
    fh.write("rocprofv3 kernel trace + PMC passes of `bench.py --smoother <sm>` (scripts/gpu_r03_profiles.sh).  VGPRs = the allocation (2 x the trace's\n"
             "VGPR_Count, which is half the compiler's .vgpr_count on this stack: checked against hipcc -S), waves/SIMD = what that allocation admits\n"
             "(MI355X_MICROARCH.md, Register files); VALU %% = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES, wait %% = SQ_WAIT_ANY /\n"
             "SQ_WAVE_CYCLES (both per wave); HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE per launch; TB/s = those bytes / average launch time.\n"
             "One instantiation runs on several levels; \"size class\" groups its dispatches by size (0 = the largest level it runs on: Op5 =\n16384^2, Op9c = 8192^2; every class is about four times fewer points; the smallest classes are merged).\nTemplate arguments of k_fused: <operator policy, smoother (0 Jacobi, 1 red-black, 2 four-colour), sweeps, flags (1 prolong, 2 restrict,\n4 zero-in, 8 no-store, 16/32 recomputed sweeps)>.\n\n")
    fh.write("| cycle | kernel | launches | avg us | VGPRs | waves/SIMD | VALU % | wait % | HBM MB/launch | TB/s |\n|---|---|---|---|---|---|---|---|---|---|\n")
    for sm, k, calls, us, vg, wps, valu, wait, hbm, rate in table_rows:
        fh.write("| %s | `%s` | %d | %.1f | %s | %s | %s | %s | %s | %s |\n" % (
            sm, k, calls, us, vg or "", wps or "", "%.0f" % valu if valu is not None else "", "%.0f" % wait if wait is not None else "",
            "%.1f" % (hbm / 1e6) if hbm else "", "%.2f" % rate if rate else ""))
print(open(os.path.join(out, "%s_kernel_table.md" % prefix)).read())
