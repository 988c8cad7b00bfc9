"""Create, step and destroy many batches (fused and rocFFT paths, with a low-resolution
observation) and report the device memory in use before and after: nothing may accumulate
except the process-wide rocFFT plans."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from scarlet_amd import synthetic  # noqa: E402
from scarlet_amd.batch import BlendBatch, ComponentSpec  # noqa: E402

kern = synthetic.psfs()
scenes = synthetic.make_batch(range(10, 42), kernel=kern)


def used():
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 2**20


def cycle(conv_path):
    comps = [[ComponentSpec(s["seds"][k], s["morphs"][k], s["origins"][k],
                            sed_min_step=s["noise_rms"]) for k in range(10)] for s in scenes]
    b = BlendBatch(np.stack([s["data"] for s in scenes]), np.stack([s["weights"] for s in scenes]),
                   comps, kernel=kern[2], max_iter=6, conv_path=conv_path)
    b.set_sub_ranges(3)
    b.step(0, 4)
    b.status()
    b.close()


for path in ("auto", "rocfft"):
    cycle(path)
    start = used()
    for _ in range(60):
        cycle(path)
    print("%-6s: %.1f MiB in use before, %.1f MiB after 60 create/step/destroy cycles"
          % (path, start, used()))
