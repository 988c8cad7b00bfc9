"""Development aid: cProfile of one `Blend.fit(100, e_rel=1e-4)` of the quickstart scene."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import golden  # noqa: E402
from test_gpu_facade import build_blend  # noqa: E402

hsc = golden("hsc_cosmos_35")
build_blend(hsc, resizing=False)[0].fit(5)
for _ in range(3):
    blend, _ = build_blend(hsc, resizing=False)
    t0 = time.perf_counter()
    blend.fit(100, e_rel=1e-4)
    print("fit %.2f ms" % ((time.perf_counter() - t0) * 1e3))
blend, _ = build_blend(hsc, resizing=False)
pr = cProfile.Profile()
pr.enable()
blend.fit(100, e_rel=1e-4)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(30)
