"""One filter step as a kernel timeline from a rocprofv3 --kernel-trace CSV (columns Kernel_Name, Start_Timestamp, End_Timestamp, Grid_Size...).
usage: python tools/step_timeline.py <kernel_trace.csv> [step_index_from_end]"""
import csv, re, sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
def short(n):
    n = re.sub(r"^void\s+", "", n)
    n = re.sub(r"pre3::", "", n)
    n = re.sub(r"\(.*$", "", n)
    return n
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Grid_Size", r.get("Grid_Size_X", "?"))) for r in rows))
starts = [i for i, e in enumerate(ev) if e[2].startswith("k_predict")]
a, b = starts[-back - 1], starts[-back]
t0 = ev[a][0]
print("# kernel | grid threads | start us | duration us | gap to previous us")
prev_end = t0
busy = 0
for s, e, n, g in ev[a:b]:
    print("%-34s %8s %8.1f %8.1f %6.1f" % (n[:34], g, (s - t0) / 1e3, (e - s) / 1e3, max(0, s - prev_end) / 1e3))
    busy += (e - s)
    prev_end = max(prev_end, e)
print("# step wall (predict_x start -> next predict_x start): %.1f us, kernel busy %.1f us" % ((ev[b][0] - t0) / 1e3, busy / 1e3))
