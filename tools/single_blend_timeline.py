"""Development aid: one blend (hsc_cosmos_35, 10 components) through the C ABI -- wall time
per iteration of smi_batch_step vs the device time of its kernels (HIP events)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch  # noqa: F401

from conftest import golden
import scarlet_amd as amd

g = golden("hsc_cosmos_35")
comps = [amd.ComponentSpec(g["sed_%d" % k], g["morph_%d" % k], g["origin_%d" % k],
                           sed_min_step=g["min_step_%d" % k]) for k in range(int(g["n_comp"]))]
for nb in (1, 8, 64):
    batch = amd.BlendBatch(np.repeat(g["images"][None], nb, 0), np.repeat(g["weights"][None], nb, 0),
                           [comps] * nb, kernel=g["diff_kernel"], max_iter=128)
    batch.step(0, 10)
    batch.status()
    t0 = time.perf_counter()
    batch.step(10, 100)
    batch.status()
    wall = (time.perf_counter() - t0) / 100
    batch.enable_timing(True)
    batch.step(110, 10)
    t = batch.timing()
    print("nb %3d: wall %.3f ms/iteration; device conv %.3f update %.3f total %.3f" % (
        nb, wall * 1e3, t["conv"], t["update"], t["total"]))
    batch.close()
