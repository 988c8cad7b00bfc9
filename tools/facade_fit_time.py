import sys, os, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from conftest import golden
from test_gpu_facade import build_blend
hsc = golden("hsc_cosmos_35")
blend, obs = build_blend(hsc, resizing=False); blend.fit(5)
for r in (False, True):
    blend, obs = build_blend(hsc, resizing=r)
    t0 = time.perf_counter(); n, logL = blend.fit(100, e_rel=1e-4); dt = time.perf_counter() - t0
    print("resizing", r, "iterations", n, "time %.1f ms" % (dt * 1e3))
