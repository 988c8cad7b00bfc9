"""Development aid: how the update kernel's time splits between the gather / AMSGrad part,
the proximal sub-iterations and, inside those, the monotonic sweep (benchmark batch,
sub-iterations capped; `--no-sweep` drops the monotonicity constraint from the chain)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (loads the HIP runtime first)

from scarlet_amd import BlendBatch, ComponentSpec, synthetic, _lib

nb = int(sys.argv[sys.argv.index("--blends") + 1]) if "--blends" in sys.argv else 1024
no_sweep = "--no-sweep" in sys.argv
flags = _lib.PROX_EXTENDED_SOURCE & ~(_lib.PROX_MONOTONIC if no_sweep else 0)
scenes = synthetic.make_batch(range(1234, 1234 + nb))
kern = synthetic.psfs()
comps = [[ComponentSpec(s["seds"][k], s["morphs"][k], s["origins"][k], sed_min_step=s["noise_rms"],
                        prox_flags=flags)
          for k in range(len(s["morphs"]))] for s in scenes]
data = np.stack([s["data"] for s in scenes])
weights = np.stack([s["weights"] for s in scenes])
for pmi in (0, 1, 2, 3, 10):
    batch = BlendBatch(data, weights, comps, kernel=kern[2], max_iter=64)
    batch.set_sub_ranges(1)
    batch.step(0, 10, e_rel=1e-3, prox_max_iter=pmi)
    batch.enable_timing(True)
    batch.step(10, 40, e_rel=1e-3, prox_max_iter=pmi)
    t = batch.timing()
    print("prox_max_iter", pmi, "update %.3f ms" % t["update"], flush=True)
    batch.close()
