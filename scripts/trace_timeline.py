"""Timeline of the LAST `count` cycles of a rocprofv3 --kernel-trace CSV: per kernel launch its stream (queue), start
offset, duration and the gap to the previous launch on the same queue; then totals (busy time per queue, wall time of the
window).  usage: trace_timeline.py <kernel_trace.csv> <marker-substring> [count] [max-lines]
The window starts at the `count`-th last launch whose name contains the marker (e.g. the cycle's first kernel)."""
import csv
import sys

path, marker = sys.argv[1], sys.argv[2]
count = int(sys.argv[3]) if len(sys.argv) > 3 else 1
max_lines = int(sys.argv[4]) if len(sys.argv) > 4 else 200
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))


def short(name):
    name = name.replace("void ", "").replace("mgcmt::fused::", "").replace("mgcmt::(anonymous namespace)::", "").replace("mgcmt::", "")
    return name.split("(")[0][:70]


marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
if len(marks) < count + 1:
    sys.exit("marker found %d times" % len(marks))
lo, hi = marks[-count - 1], marks[-1]
win = rows[lo:hi]
t0 = int(win[0]["Start_Timestamp"])
last_end = {}
busy = {}
print("%-70s %6s %10s %9s %9s" % ("kernel", "queue", "start_us", "dur_us", "gap_us"))
for n, r in enumerate(win):
    q = r.get("Queue_Id", "?")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    busy[q] = busy.get(q, 0.0) + (e - s) / 1e3
    if n < max_lines:
        print("%-70s %6s %10.1f %9.1f %9.1f" % (short(r["Kernel_Name"]), q, (s - t0) / 1e3, (e - s) / 1e3, gap))
wall = (int(rows[hi]["Start_Timestamp"]) - t0) / 1e3
print("window: %d launches, %.1f us wall (%d cycle(s): %.1f us each)" % (len(win), wall, count, wall / count))
for q, b in busy.items():
    print("queue %s busy %.1f us (%.1f per cycle)" % (q, b, b / count))
