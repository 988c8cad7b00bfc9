import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from conftest import golden
from test_gpu_facade import build_blend
hsc = golden("hsc_cosmos_35")
blend, obs = build_blend(hsc, resizing=False)
blend.fit(5)  # warm-up (library load, rocFFT kernels)
blend, obs = build_blend(hsc, resizing=False)
pr = cProfile.Profile(); pr.enable()
n, logL = blend.fit(100, e_rel=1e-4)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
