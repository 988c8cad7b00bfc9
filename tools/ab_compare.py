"""Development aid: run the same fit with two builds of libscarlet_amd.so (the working
tree's and a reference copy, e.g. the previous round's) in separate processes and compare
losses, parameters and moments bit for bit.

    python tools/ab_compare.py [--config cfg3|cfg1] [--blends 64] [--steps 30] [--ref tools/ab/libscarlet_amd_ref.so]
"""
import argparse
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(args):
    import bench
    from scarlet_amd import BlendBatch

    if args.config == "cfg1":
        data, weights, comps, kernel = bench.build_cfg1(args.blends)
    else:
        data, weights, comps, kernel, _ = bench.build_cfg3(0, args.blends, 0, None)
    batch = BlendBatch(data, weights, comps, kernel=kernel, max_iter=args.steps + 1)
    batch.set_sub_ranges(args.sub_ranges)
    batch.step(0, args.steps, e_rel=1e-3, check_convergence=bool(args.check))
    out = dict(loss=np.concatenate(batch.loss_history()))
    seds, morphs = batch.parameters()
    out["seds"] = seds
    out["morphs"] = np.concatenate([m.reshape(-1) for m in morphs])
    mom = batch.moments()
    for k in ("m_sed", "v_sed", "vhat_sed"):
        out[k] = mom[k]
    for k in ("m_morph", "v_morph", "vhat_morph"):
        out[k] = np.concatenate([m.reshape(-1) for m in mom[k]])
    np.savez(args.out, **out)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--blends", type=int, default=64)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--sub-ranges", type=int, default=0)
    ap.add_argument("--check", type=int, default=0)
    ap.add_argument("--ref", default=os.path.join(ROOT, "tools", "ab", "libscarlet_amd_ref.so"))
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    if args.out:
        run(args)
        sys.exit(0)
    outs = []
    for tag, lib in (("new", None), ("ref", args.ref)):
        env = dict(os.environ)
        if lib:
            env["SCARLET_AMD_LIB"] = lib
        out = "/tmp/ab_%s.npz" % tag
        subprocess.check_call([sys.executable, __file__, "--out", out] + sys.argv[1:], env=env)
        outs.append(np.load(out))
    worst = 0.0
    same = True
    for k in outs[0].files:
        a, b = outs[0][k], outs[1][k]
        eq = np.array_equal(a, b, equal_nan=True)
        same &= eq
        d = float(np.nanmax(np.abs(a - b) / (np.abs(b) + 1e-30))) if a.size else 0.0
        worst = max(worst, d)
        print("%-12s %s  max rel diff %.3g" % (k, "identical" if eq else "DIFFERENT", d))
    print("AB:", "bit-identical" if same else "differs (worst %.3g)" % worst)
