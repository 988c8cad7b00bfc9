"""Reduce the rocprofv3 --pmc passes of tools/collect_profiles.sh to per-kernel figures.

    python tools/hbm_counters.py <gpurun_out/tag directory> <tag> <blends per launch>

Writes <dir>/<tag>_counters.csv (mean per launch of every counter, full-batch launches
only) and <dir>/hbm_traffic.json, which bench.py reads from profiles/ for the `roofline`
object: HBM bytes per blend (`traffic`), VALU utilisation and what binds the kernel.

Corrections as MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE and WRITE_SIZE are
reported in KiB; FETCH_SIZE counts 64 B per 128-B request -> doubled; WRITE_SIZE as
reported.  SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES count quad-cycles summed over the SIMDs:
utilisation of a pipe = 4 x counter / (kernel cycles x 1024 SIMDs); GRBM_GUI_ACTIVE is
summed over the 8 XCDs, so kernel cycles = GRBM_GUI_ACTIVE / 8 (0.754 ms x 2.41 GHz for
the convolution kernel, consistent with the kernel trace).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out_dir, tag, nb = sys.argv[1], sys.argv[2], int(sys.argv[3])
N_SIMD = 256 * 4


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("smi::", "")
    return name.split("(")[0]


vals = defaultdict(lambda: defaultdict(list))  # kernel -> counter -> per-launch values
for path in glob.glob(os.path.join(out_dir, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        vals[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))


def full(v):
    """mean over the full-batch launches (the upper half: warm-up batches are smaller)"""
    v = sorted(v)[len(v) // 2:]
    return sum(v) / len(v)


lines = ["kernel,counter,launches,mean_per_launch"]
for k in sorted(vals):
    for c in sorted(vals[k]):
        lines.append('"%s",%s,%d,%.1f' % (k, c, len(vals[k][c]), full(vals[k][c])))
open(os.path.join(out_dir, tag + "_counters.csv"), "w").write("\n".join(lines) + "\n")

out = {}
for key in ("fused_conv_kernel", "update_kernel_reg", "render_kernel"):
    # (a kernel that only runs a few times -- render_kernel draws the synthetic scenes' data --
    # is no part of the iteration)
    match = [k for k in vals if k.startswith(key) and max(len(v) for v in vals[k].values()) >= 10]
    if not match:
        continue
    c = {name: full(v) for k in match for name, v in vals[k].items()}
    rec = {"note": "%d-blend launches (%s); rocprofv3 --pmc, separate passes" % (nb, tag)}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        rec["bytes_per_blend"] = int(round((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 / nb))
        rec["fetch_KiB_reported"] = c["FETCH_SIZE"]
        rec["write_KiB_reported"] = c["WRITE_SIZE"]
    if "GRBM_GUI_ACTIVE" in c:
        cycles = c["GRBM_GUI_ACTIVE"] / 8  # summed over the XCDs
        rec["kernel_cycles"] = cycles
        for name, field in (("SQ_ACTIVE_INST_VALU", "valu_busy"), ("SQ_ACTIVE_INST_ANY", "issue_busy")):
            if name in c:
                rec[field] = round(4 * c[name] / (cycles * N_SIMD), 4)
        if "SQ_WAVE_CYCLES" in c:
            rec["waves_per_simd"] = round(4 * c["SQ_WAVE_CYCLES"] / (cycles * N_SIMD), 2)
    if c.get("SQ_LDS_IDX_ACTIVE") and "SQ_LDS_BANK_CONFLICT" in c:
        rec["lds_conflict_frac"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 4)
    # what binds the kernel: HBM if the measured traffic needs most of its time at 8 TB/s,
    # else instruction issue (the VALU pipe is the busiest one in both kernels)
    if "bytes_per_blend" in rec and "kernel_cycles" in rec:
        hbm_frac = rec["bytes_per_blend"] * nb / (rec["kernel_cycles"] / 2.4e9) / 8e12
        rec["hbm_frac_measured_at_2.4GHz"] = round(hbm_frac, 4)
        # HBM if the measured traffic needs most of the time at 8 TB/s; instruction issue if the
        # SIMDs issue in most cycles; otherwise the kernel waits -- for the update kernel on the
        # plan stream out of L2 and the LDS round trip of a sweep step
        rec["bound"] = ("hbm" if hbm_frac > 0.6 else
                        "valu-issue" if rec.get("issue_busy", 0) > 0.85 else "latency-l2-lds")
    out[key] = rec
json.dump(out, open(os.path.join(out_dir, "hbm_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
