"""Turn one run of scripts/gpu_round1_final.sh (gpurun_out/*_<tag>*) into the tracked summaries under profiles/:
HBM bytes per launch of the dominant kernel from the two PMC passes (FETCH_SIZE doubled on gfx950 as
MI355X_MICROARCH.md prescribes for 16-byte streams; counter unit KB), the rocprofv3 kernel stats, the bench lines.
usage: collect_profiles.py <tag> <out-prefix>      e.g.  collect_profiles.py r01g r01"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, prefix = sys.argv[1], sys.argv[2]
out = os.path.join(ROOT, "profiles")
g = os.path.join(ROOT, "gpurun_out")
KERNEL = {"wjacobi": "k_fused<mgcmt::fused::Op5, 0, 2, 0>", "rb": "k_fused<mgcmt::fused::Op5, 1, 2, 0>"}
n = 16384
detail, table = {}, {}
for sm, kname in KERNEL.items():
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        files = glob.glob(os.path.join(g, "pmc_%s_%s_%s" % (tag, sm, counter), "**", "*counter_collection.csv"), recursive=True)
        files = sorted(files, key=os.path.getmtime)[-1:]          # the latest run only (a tag may have been used before)
        acc = []
        for f in files:
            for row in csv.DictReader(open(f)):
                if kname in row["Kernel_Name"] and row["Counter_Name"] == counter:
                    acc.append(float(row["Counter_Value"]))
        vals[counter] = (sum(acc) / len(acc), len(acc)) if acc else (None, 0)
    if vals["FETCH_SIZE"][0] is None or vals["WRITE_SIZE"][0] is None:
        continue
    fetch = vals["FETCH_SIZE"][0] * 1024 * 2
    write = vals["WRITE_SIZE"][0] * 1024
    detail["%s_%d" % (sm, n)] = {
        "kernel": kname, "FETCH_SIZE_KB_avg": vals["FETCH_SIZE"][0], "WRITE_SIZE_KB_avg": vals["WRITE_SIZE"][0],
        "dispatches": vals["FETCH_SIZE"][1], "fetch_bytes_corrected_x2": fetch, "write_bytes": write,
        "algorithmic_bytes_24B_per_update": 24 * n * n * 2, "bytes_one_pass_reads_v_f_writes_v": 24 * n * n,
        "note": "one launch = 2 sweeps; HBM bytes per launch; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 16-byte streams at half)"}
    table["%s_%d" % (sm, n)] = fetch + write
# the other fine-level kernels of the cycle (same PMC passes): HBM bytes per launch next to their algorithmic bytes
OTHER = {"wjacobi": {"k_fused<mgcmt::fused::Op5, 0, 2, 10>": ("down pass: 2 sweeps + residual + restriction, V' not stored", 16 + 2),
                     "k_fused<mgcmt::fused::Op5, 0, 2, 33>": ("up pass: 2 recomputed sweeps + correction + 2 sweeps", 24 + 2)},
         "rb": {"k_fused<mgcmt::fused::Op5, 1, 2, 10>": ("down pass: 2 sweeps + residual + restriction, V' not stored", 16 + 2),
                "k_fused<mgcmt::fused::Op5, 1, 2, 33>": ("up pass: 2 recomputed sweeps + correction + 2 sweeps", 24 + 2)}}
for sm, kernels in OTHER.items():
    for kname, (what, bpp) in kernels.items():
        vals = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            files = sorted(glob.glob(os.path.join(g, "pmc_%s_%s_%s" % (tag, sm, counter), "**", "*counter_collection.csv"), recursive=True),
                           key=os.path.getmtime)[-1:]
            acc = [float(r["Counter_Value"]) for f in files for r in csv.DictReader(open(f))
                   if kname in r["Kernel_Name"] and r["Counter_Name"] == counter]
            vals[counter] = sum(acc) / len(acc) if acc else None
        if vals["FETCH_SIZE"] is not None and vals["WRITE_SIZE"] is not None:
            detail["%s_%d %s" % (sm, n, kname)] = {"what": what, "fetch_bytes_corrected_x2": vals["FETCH_SIZE"] * 2048,
                                                  "write_bytes": vals["WRITE_SIZE"] * 1024, "algorithmic_bytes": bpp * n * n}
if table:
    json.dump(table, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
    json.dump(detail, open(os.path.join(out, "%s_pmc_traffic_detail.json" % prefix), "w"), indent=1)
for sm in KERNEL:
    stats = sorted(glob.glob(os.path.join(g, "prof_%s_%s" % (tag, sm), "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    if stats:
        shutil.copy(stats[-1], os.path.join(out, "%s_final_%s_%d_kernel_stats.csv" % (prefix, sm, n)))
for src, dst in (("bench_%s_default.json" % tag, "%s_bench_default_16384_wjacobi.json" % prefix), ("bench_%s_rb.json" % tag, "%s_bench_16384_rb.json" % prefix)):
    if os.path.exists(os.path.join(g, src)):
        shutil.copy(os.path.join(g, src), os.path.join(out, dst))
# the bench line printed INSIDE the profiled run (same process as the kernel stats above)
for sm in KERNEL:
    log = os.path.join(g, "prof_%s_%s.log" % (tag, sm))
    if os.path.exists(log):
        lines = [l for l in open(log) if l.startswith('{"metric"')]
        if lines:
            open(os.path.join(out, "%s_bench_under_rocprof_%s.json" % (prefix, sm)), "w").write(lines[-1])
print(json.dumps(detail, indent=1))
