import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from scarlet_amd import BlendBatch, ComponentSpec, synthetic
nb = 1024
kern = synthetic.psfs()
scenes = synthetic.make_batch(range(1234, 1234 + nb), kernel=kern)
comps = [[ComponentSpec(s["seds"][k], s["morphs"][k], s["origins"][k], sed_min_step=s["noise_rms"]) for k in range(10)] for s in scenes]
data = np.stack([s["data"] for s in scenes]); weights = np.stack([s["weights"] for s in scenes])
pr = cProfile.Profile(); pr.enable()
b = BlendBatch(data, weights, comps, kernel=kern[2], max_iter=101)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
b.close()
# single-blend facade fit latency (cfg 1)
from conftest import golden, hsc_scene
from test_gpu_facade import build_blend
hsc = golden("hsc_cosmos_35")
for rep in range(3):
    blend, obs = build_blend(hsc, resizing=False)
    t0 = time.perf_counter(); n, logL = blend.fit(100, e_rel=1e-4); t1 = time.perf_counter()
    print("Blend.fit hsc_cosmos_35: %d iterations in %.1f ms (%.2f ms/iteration)" % (n, 1e3 * (t1 - t0), 1e3 * (t1 - t0) / n))
sc = hsc_scene(hsc)
t0 = time.perf_counter(); n, logL = sc.fit(100, e_rel=1e-4); t1 = time.perf_counter()
print("oracle (CPU, 1 thread) same fit: %d iterations in %.1f ms" % (n, 1e3 * (t1 - t0)))
