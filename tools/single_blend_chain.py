"""Development aid: the quickstart blend (hsc_cosmos_35) through the C ABI, iterations 0 .. 75 of
a fresh fit (the expensive early iterations included): wall and device time per iteration."""
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
for nb in (1, 16):
    batch = amd.BlendBatch(np.repeat(g["images"][None], nb, 0), np.repeat(g["weights"][None], nb, 0),
                           [comps] * nb, kernel=g["diff_kernel"], max_iter=128)
    batch.save_state()
    batch.step(0, 76)
    batch.status()
    best = 1e9
    for _ in range(5):
        batch.restore_state()
        batch.status()
        t0 = time.perf_counter()
        batch.step(0, 76)
        batch.status()
        best = min(best, (time.perf_counter() - t0) / 76)
    batch.restore_state()
    batch.enable_timing(True)
    batch.step(0, 76)
    t = batch.timing()
    print("nb %3d: wall %.4f ms/iteration; device conv %.4f update %.4f total %.4f" % (
        nb, best * 1e3, t["conv"], t["update"], t["total"]))
    batch.close()
