"""Worker of tests/test_gpu_dist.py: one rank of a multi-rank fit.  Started through
``scarlet_amd.dist.launch_command`` (torch.distributed.run); with
SCARLET_AMD_SHARE_GPU=1 all ranks use GPU 0 and talk over gloo, which is how a
single-GPU box exercises the N > 1 path.

    dist_worker.py batch  <out.npz>   sharded BlendBatch fit of synthetic blends (C ABI)
    dist_worker.py facade <out.npz>   fit_blends(devices="ranks") of quickstart blends
"""

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

N_BLENDS, MAX_ITER, E_REL = 11, 40, 1e-3   # ragged over 2 ranks: 6 + 5


def make_batch_factory(device):
    from scarlet_amd import BlendBatch, ComponentSpec, synthetic

    kern = synthetic.psfs()

    def make_batch(lo, hi):
        # fewer sources in odd blends, so that the blends stop at different iterations
        scenes = [synthetic.make_blend(1234 + b, kernel=kern, n_sources=10 - 3 * (b % 2))
                  for b in range(lo, hi)]
        comps = [[ComponentSpec(s["seds"][k], s["morphs"][k], s["origins"][k],
                                sed_min_step=s["noise_rms"]) for k in range(len(s["morphs"]))]
                 for s in scenes]
        return BlendBatch(np.stack([s["data"] for s in scenes]),
                          np.stack([s["weights"] for s in scenes]), comps, kernel=kern[2],
                          max_iter=MAX_ITER, device=device)

    return make_batch


def run_batch(device):
    from scarlet_amd import dist

    rec, (seds, morphs) = dist.fit_sharded(make_batch_factory(device), N_BLENDS, max_iter=MAX_ITER,
                                           e_rel=E_REL, with_parameters=True)
    return dict(n_iter=rec["n_iter"], converged=rec["converged"], logL=rec["logL"],
                loss_hist=rec["loss_hist"], seds=seds,
                morphs=np.concatenate([m.reshape(-1) for m in morphs]))


def facade_blends():
    from conftest import golden
    from test_gpu_facade import build_blend
    import scarlet_amd as scarlet

    hsc = golden("hsc_cosmos_35")
    blends = []
    for k in range(3):
        full, obs = build_blend(hsc, resizing=True)
        sources = list(full.sources)
        blend = scarlet.Blend(sources[:len(sources) - k], obs)
        for p in blend.parameters:
            if p.name == "spectrum":
                p *= 1 + 0.1 * k
        blends.append(blend)
    return blends


def facade_summary(blends, results):
    out = dict(results=np.array(results, dtype=np.float64))
    for i, blend in enumerate(blends):
        out["loss_%d" % i] = np.array(blend.loss)
        out["bbox_%d" % i] = np.array([list(s.bbox.origin) + list(s.bbox.shape)
                                       for s in blend.sources])
        for j, p in enumerate(blend.parameters):
            out["p_%d_%d" % (i, j)] = np.asarray(p)
            if p.m is not None:
                out["m_%d_%d" % (i, j)] = np.asarray(p.m)
                out["std_%d_%d" % (i, j)] = np.ma.filled(p.std, 0.0)
    return out


def main():
    mode, out_path = sys.argv[1], sys.argv[2]
    import torch
    from scarlet_amd import dist

    rank, local_rank, world = dist.env_rank()
    if os.environ.get("SCARLET_AMD_SHARE_GPU") == "1":
        os.environ["LOCAL_RANK"] = str(local_rank % torch.cuda.device_count())
        local_rank = int(os.environ["LOCAL_RANK"])
    dist.init_process_group(backend=os.environ.get("SCARLET_AMD_DIST_BACKEND"),
                            device_index=local_rank)
    if mode == "batch":
        out = run_batch(local_rank)
    else:
        import scarlet_amd as scarlet

        blends = facade_blends()
        results = scarlet.fit_blends(blends, 35, e_rel=1e-5, devices="ranks")
        out = facade_summary(blends, results)
    # every rank holds the whole job's results: all ranks write, the test compares them
    np.savez(out_path.replace(".npz", "_rank%d.npz" % rank), **out)
    dist.barrier()


if __name__ == "__main__":
    main()
