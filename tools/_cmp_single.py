import copy, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import golden
from test_gpu_facade import build_blend
import bench
import scarlet_amd as scarlet

def med(f, make, repeat=7):
    f(make())
    out = []
    for _ in range(repeat):
        b = make(); t0 = time.perf_counter(); r = f(b); out.append((time.perf_counter() - t0, r))
    out.sort(key=lambda x: x[0]); return out[len(out)//2]
hsc = golden("hsc_cosmos_35")
blends = bench.build_facade_blends(0, 1, 0)
for name, make in (("quickstart resizing on", lambda: build_blend(hsc, resizing=True)[0]),
                   ("quickstart resizing off", lambda: build_blend(hsc, resizing=False)[0]),
                   ("configs[1] scene", lambda: copy.deepcopy(blends[0]))):
    a = med(lambda b: b.fit(100, e_rel=1e-4), make)
    c = med(lambda b: scarlet.fit_blends([b], 100, e_rel=1e-4)[0], make)
    print(name, "Blend.fit %.2f ms %s | fit_blends([b]) %.2f ms %s" % (a[0]*1e3, a[1], c[0]*1e3, c[1]))
