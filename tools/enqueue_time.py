"""Development aid: how long the host takes to ENQUEUE the iterations of a fit next to how
long the device takes to run them (a small shard is twenty-odd launches per iteration).
    python tools/enqueue_time.py [blends] [iterations]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from scarlet_amd import BlendBatch, ComponentSpec, synthetic

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 128
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
scenes = synthetic.make_batch(range(1234, 1234 + nb))
kern = synthetic.psfs()
comps = [[ComponentSpec(s["seds"][k], s["morphs"][k], s["origins"][k], sed_min_step=s["noise_rms"])
          for k in range(len(s["morphs"]))] for s in scenes]
b = BlendBatch(np.stack([s["data"] for s in scenes]), np.stack([s["weights"] for s in scenes]), comps,
               kernel=kern[2], max_iter=4 * K + 8)
b.step(0, 5)
b.status()
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    b.step(5 + rep * K, K)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("blends %d ranges %d: enqueue %.3f ms, until done %.3f ms (%d iterations): %.1f us of host per iteration, %.1f us of device"
          % (nb, b.sub_ranges(), (t1 - t0) * 1e3, (t2 - t0) * 1e3, K, (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6))
