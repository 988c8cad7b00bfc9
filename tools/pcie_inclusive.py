"""PCIe-inclusive rate: host buffers -> device (data, weights, parameters) + 100 iterations +
results back, for 1024 blends.  Not the benchmark value."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scarlet_amd import BlendBatch, ComponentSpec, synthetic
nb, iters = 1024, 100
kern = synthetic.psfs()
scenes = synthetic.make_batch(range(1234, 1234 + nb), kernel=kern)
comps = [[ComponentSpec(s["seds"][k], s["morphs"][k], s["origins"][k], sed_min_step=s["noise_rms"]) for k in range(10)] for s in scenes]
data = np.stack([s["data"] for s in scenes]); weights = np.stack([s["weights"] for s in scenes])
for rep in range(2):
    t0 = time.perf_counter()
    b = BlendBatch(data, weights, comps, kernel=kern[2], max_iter=iters + 1)
    t1 = time.perf_counter()
    b.step(0, iters); b.status()
    t2 = time.perf_counter()
    sed, morphs = b.parameters(); loss = b.loss_history()
    t3 = time.perf_counter()
    b.close()
    print("rep", rep, "setup+H2D %.3f s, %d iterations %.3f s, D2H %.3f s -> %.0f blend-it/s PCIe-inclusive (setup includes Python packing of 10240 components)" % (t1 - t0, iters, t2 - t1, t3 - t2, nb * iters / (t3 - t0)))
