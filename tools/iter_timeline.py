"""Development aid: start / end of every kernel of a few iterations from a rocprofv3 kernel
trace (csv):  python tools/iter_timeline.py <dir> [first row] [rows]"""
import csv, glob, sys
rows = []
for p in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
i0 = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
n = int(sys.argv[3]) if len(sys.argv) > 3 else 24
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = t0
for r in rows[i0:i0 + n]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("smi::(anonymous namespace)::", "").replace("void ", "")[:34]
    print("%-34s q%-3s start %8.1f us  dur %6.1f  gap after previous end %6.1f" % (
        name, r.get("Queue_Id", "?"), (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
    prev_end = max(prev_end, e)
