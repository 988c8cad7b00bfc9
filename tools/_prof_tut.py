"""Per-C-call wall time of tools/multires_tutorial.py (development aid)."""
import collections
import os
import runpy
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from scarlet_amd import _lib  # noqa: E402

lib = _lib.load()
acc = collections.defaultdict(lambda: [0, 0.0])


class Timed:
    def __init__(self, name, fn):
        self.name, self.fn = name, fn

    def __call__(self, *a):
        t = time.perf_counter()
        r = self.fn(*a)
        e = acc[self.name]
        e[0] += 1
        e[1] += time.perf_counter() - t
        return r


for name in _lib.SYMBOLS:
    setattr(lib, name, Timed(name, getattr(lib, name)))
sys.argv = ["multires_tutorial.py", "30"]
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "multires_tutorial.py"),
               run_name="__main__")
for name, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:14]:
    print("%-40s %5d calls %8.3f s" % (name, n, t))
