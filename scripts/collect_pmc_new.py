"""Per-launch HBM bytes of the round-3 kernels from the PMC passes of scripts/gpu_r03_pmc_new.sh: the largest dispatches of
each kernel (the finest level), FETCH_SIZE x 2 (gfx950 tallies the 128-byte requests of 16-byte streams at 64 B,
MI355X_MICROARCH.md "HBM") + WRITE_SIZE, counters in KB.  Writes profiles/r03_pmc_new_kernels.json."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = os.path.join(ROOT, "gpurun_out")
out = {}
for what, names in (("1d", ("k_fused1d",)), ("rqmg", ("k_rq_pass1<6, 0>", "k_rq_pass2<6, 0>", "k_rq_pass1<3, 1>", "k_rq_pass2<3, 1>", "k_rq_gmg"))):
    per = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        files = glob.glob(os.path.join(g, "pmc_r03_%s_%s" % (what, counter), "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        for r in csv.DictReader(open(files[0])):
            if r["Counter_Name"] != counter:
                continue
            for nm in names:
                if nm in r["Kernel_Name"]:
                    per.setdefault((nm, r["Kernel_Name"]), {}).setdefault(counter, []).append(float(r["Counter_Value"]))
    for (nm, full), c in per.items():
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            continue
        top_f = max(c["FETCH_SIZE"])
        top_w = max(c["WRITE_SIZE"])
        big_f = [v for v in c["FETCH_SIZE"] if v > 0.7 * top_f]
        big_w = [v for v in c["WRITE_SIZE"] if v > 0.7 * top_w] if top_w > 0 else [0.0]
        key = full.replace("void ", "").replace("mgcmt::(anonymous namespace)::", "").replace("mgcmt::fused1d::", "").split("(")[0]
        out["%s: %s" % (what, key)] = {"launches_of_the_largest_size": len(big_f), "fetch_bytes_corrected_x2": 2048 * sum(big_f) / len(big_f),
                                        "write_bytes": 1024 * sum(big_w) / len(big_w),
                                        "hbm_bytes": 2048 * sum(big_f) / len(big_f) + 1024 * sum(big_w) / len(big_w)}
json.dump(out, open(os.path.join(ROOT, "profiles", "r03_pmc_new_kernels.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
