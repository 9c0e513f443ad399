"""One cycle of the emulated-rank run out of a rocprofv3 kernel trace: the last TIMED cycle (before the k_fill that starts
the residual-check phase).  usage: cycle_timeline.py <kernel_trace.csv> [marker]"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2] if len(sys.argv) > 2 else "Op5, 1, 2, 10"


def short(name):
    name = name.replace("void ", "").replace("mgcmt::fused::", "").replace("mgcmt::(anonymous namespace)::", "").replace("mgcmt::", "")
    return name.split("(")[0][:60]


fills = [i for i, r in enumerate(rows) if "k_fill" in r["Kernel_Name"]]
end = fills[-1]
sel = rows[max(0, end - 200):end]
# a cycle starts at the first marker launch after a launch of another kernel family (the up pass of the previous cycle)
idx = [i for i, r in enumerate(sel) if marker in r["Kernel_Name"] and (i == 0 or "Op5, 1, 2, 33" in sel[i - 1]["Kernel_Name"] or "rccl" in sel[i - 1]["Kernel_Name"] and "Op5, 1, 2, 33" in sel[i - 2]["Kernel_Name"])]
start, stop = idx[-2], idx[-1]
t0 = int(sel[start]["Start_Timestamp"])
prev = {}
print("%-60s %5s %6s %10s %9s %8s" % ("kernel", "queue", "stream", "start_us", "dur_us", "gap_us"))
for r in sel[start:stop]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = r["Queue_Id"]
    gap = (s - prev.get(q, s)) / 1e3
    prev[q] = e
    print("%-60s %5s %6s %10.1f %9.1f %8.1f" % (short(r["Kernel_Name"]), q, r["Stream_Id"], (s - t0) / 1e3, (e - s) / 1e3, gap))
print("cycle: %.1f us" % ((int(sel[stop]["Start_Timestamp"]) - t0) / 1e3))
