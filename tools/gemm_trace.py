"""Development aid: per-dispatch durations of the resampler's kernels from a rocprofv3 kernel
trace (csv), grouped by grid size:  python tools/gemm_trace.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys, collections
rows = []
for p in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(p)))
g = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "gemm_" in n or "reduce_slices" in n:
        key = (n.replace("smi::(anonymous namespace)::", "").replace("void ", "").split("(")[0],
               r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r.get("VGPR_Count"), r.get("Accum_VGPR_Count"))
        g[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
    print(k, "calls", len(v), "avg us %.1f" % (sum(v) / len(v) / 1e3), "min %.1f" % (min(v) / 1e3))
