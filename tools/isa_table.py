"""ISA resource table of the device kernels: VGPRs, AGPRs, SGPR spills, scratch bytes, LDS, and the
instruction mix (VALU / SALU / DS / VMEM / scratch ops, packed and DPP) per kernel instance,
from the assembly `hipcc -save-temps` leaves behind.

    python tools/isa_table.py [kernels.hip ...] [--filter update_kernel] [--out profiles/r06_isa.txt]
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "scarlet_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-Wall",
         "-Wno-unused-function"]
FFT_FLAGS = ["-ffp-contract=fast", "-fno-slp-vectorize", "-fno-signed-zeros"]


def demangle(names):
    out = subprocess.run(["c++filt"] + names, capture_output=True, text=True)
    return out.stdout.split("\n")[:len(names)]


def assembly(src, tmp, extra=()):
    flags = FLAGS + (FFT_FLAGS if "fused_conv" in src else []) + list(extra)
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-save-temps", "-c", os.path.join(CSRC, src),
                                                             "-o", os.path.join(tmp, "x.o")], cwd=tmp,
                          stderr=subprocess.DEVNULL)
    stem = os.path.splitext(src)[0]
    return open(os.path.join(tmp, stem + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read()


def parse(text):
    """{mangled name: dict} from the .amdhsa_ directives and the body of every kernel."""
    rows = {}
    # bodies: from "<name>:" to ".Lfunc_end"
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        ins = [l.strip().split()[0] for l in body.split("\n")
               if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        mix = dict(valu=0, salu=0, ds=0, vmem=0, scratch=0, pk=0, dpp=0, total=len(ins))
        for l in body.split("\n"):
            if "row_ror" in l or "row_shr" in l or "quad_perm" in l or "row_bcast" in l or "row_newbcast" in l:
                mix["dpp"] += 1
        for i in ins:
            if i.startswith("v_"):
                mix["valu"] += 1
                mix["pk"] += i.startswith("v_pk_")
            elif i.startswith("s_"):
                mix["salu"] += 1
            elif i.startswith("ds_"):
                mix["ds"] += 1
            elif i.startswith("scratch_"):
                mix["scratch"] += 1
            elif i.startswith(("buffer_", "global_", "flat_")):
                mix["vmem"] += 1
        rows[name] = mix
    for m in re.finditer(r"\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel", text, re.S):
        name, d = m.group(1), m.group(2)
        r = rows.setdefault(name, {})
        for key, tag in (("lds", "group_segment_fixed_size"), ("scratch_bytes", "private_segment_fixed_size"),
                         ("vgpr_next", "next_free_vgpr"), ("accum_offset", "accum_offset")):
            mm = re.search(r"\.amdhsa_%s (\d+)" % tag, d)
            r[key] = int(mm.group(1)) if mm else None
    for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)(?=\n  - |\namdhsa\.target)", text, re.S):
        pass
    # metadata block: sgpr / vgpr spill counts
    for m in re.finditer(r"- \.agpr_count:\s+(\d+)(.*?)\.name:\s+(\S+)(.*?)\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)",
                         text, re.S):
        name = m.group(3)
        blk = m.group(2) + m.group(4)
        r = rows.setdefault(name, {})
        r["agpr"] = int(m.group(1))
        r["vgpr"] = int(m.group(5))
        r["vgpr_spill"] = int(m.group(6))
        mm = re.search(r"\.sgpr_spill_count:\s+(\d+)", blk)
        r["sgpr_spill"] = int(mm.group(1)) if mm else None
        mm = re.search(r"\.sgpr_count:\s+(\d+)", blk)
        r["sgpr"] = int(mm.group(1)) if mm else None
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("sources", nargs="*", default=["kernels.hip", "fused_conv.hip"])
    ap.add_argument("--filter", default="")
    ap.add_argument("--out")
    ap.add_argument("--extra", default="", help="extra compiler flags (quoted)")
    args = ap.parse_args()
    lines = ["%-64s %5s %5s %6s %6s %7s %7s | %6s %6s %5s %5s %5s %5s %5s" % (
        "kernel", "vgpr", "agpr", "vspill", "sspill", "scratch", "lds", "valu", "salu", "ds", "vmem",
        "scr", "pk", "dpp")]
    for src in args.sources:
        with tempfile.TemporaryDirectory() as tmp:
            rows = parse(assembly(src, tmp, args.extra.split()))
        names = sorted(n for n in rows if "vgpr" in rows[n])
        for n, pretty in zip(names, demangle(names)):
            pretty = re.sub(r"^void ", "", pretty).replace("smi::", "").replace("(anonymous namespace)::", "")
            pretty = re.sub(r"\(.*$", "", pretty)
            if args.filter and args.filter not in pretty:
                continue
            r = rows[n]
            lines.append("%-64s %5s %5s %6s %6s %7s %7s | %6s %6s %5s %5s %5s %5s %5s" % (
                pretty[:64], r.get("vgpr"), r.get("agpr"), r.get("vgpr_spill"), r.get("sgpr_spill"),
                r.get("scratch_bytes"), r.get("lds"), r.get("valu"), r.get("salu"), r.get("ds"),
                r.get("vmem"), r.get("scratch"), r.get("pk"), r.get("dpp")))
    text = "\n".join(lines)
    print(text)
    if args.out:
        with open(args.out, "w") as fh:
            fh.write(text + "\n")


if __name__ == "__main__":
    main()
