import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from scarlet_amd import _lib
lib = _lib.load()
orig = {}
acc = {}
def wrap(name):
    fn = getattr(lib, name)
    def w(*a):
        t = time.perf_counter(); r = fn(*a); acc[name] = acc.get(name, 0) + time.perf_counter() - t; return r
    return w
class Proxy:
    def __getattr__(self, n):
        return wrap(n) if n.startswith("smi_batch") else getattr(lib, n)
from scarlet_amd import batch as B
from conftest import golden
from test_gpu_facade import build_blend
hsc = golden("hsc_cosmos_35")
blend, obs = build_blend(hsc, resizing=False)
blend.fit(3)
_lib._lib = Proxy()
for rep in range(2):
    acc.clear()
    blend, obs = build_blend(hsc, resizing=False)
    t0 = time.perf_counter(); blend.fit(30); t1 = time.perf_counter()
    print("fit(30): %.1f ms" % (1e3 * (t1 - t0)), {k: round(1e3 * v, 1) for k, v in sorted(acc.items(), key=lambda kv: -kv[1])})
