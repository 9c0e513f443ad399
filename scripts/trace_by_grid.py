"""A rocprofv3 --kernel-trace CSV summed per (kernel, grid size): launches, total and mean duration — which LEVEL of a
cycle the time of a kernel belongs to (the per-kernel stats table adds all levels up) — and the idle time between
launches.  usage: trace_by_grid.py <kernel_trace.csv> [divide-by (e.g. the number of cycles traced)]"""
import csv
import sys

path = sys.argv[1]
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))


def short(name):
    name = name.replace("void ", "").replace("mgcmt::fused::", "").replace("mgcmt::(anonymous namespace)::", "").replace("mgcmt::", "")
    return name.split("(")[0][:60]


acc = {}
for r in rows:
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])  # (threads)
    key = (short(r["Kernel_Name"]), grid, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
    a = acc.setdefault(key, [0, 0.0])
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
busy = sum(a[1] for a in acc.values())
wall = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
print("%-60s %10s %5s %8s %10s %8s" % ("kernel", "grid", "wg", "launches", "total_us", "mean_us"))
for (name, grid, wg), (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print("%-60s %10d %5d %8.1f %10.1f %8.2f" % (name, grid, wg, n / div, t / div, t / n))
print("busy %.1f us, first launch to last end %.1f us (per unit: %.1f / %.1f)" % (busy, wall, busy / div, wall / div))
