"""Latency of ONE scene through the Python API (BASELINE configs[0] and configs[1]): wall time of
`Blend.fit(100, e_rel=1e-4)` on the quickstart blend (hsc_cosmos_35, boxes 21^2 .. 61^2) and on
one synthetic 5-band 128x128 scene with 10 ExtendedSource-like components (41^2), after a
warm-up fit.  One JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from conftest import golden  # noqa: E402
from test_gpu_facade import build_blend  # noqa: E402
import bench  # noqa: E402


def timed_fit(make, repeat=5):
    make().fit(5)  # warm-up (library load, plan tables)
    out = []
    for _ in range(repeat):
        blend = make()
        t0 = time.perf_counter()
        n, logL = blend.fit(100, e_rel=1e-4)
        out.append((time.perf_counter() - t0, n, float(logL)))
    out.sort()
    return out[len(out) // 2]


def main():
    hsc = golden("hsc_cosmos_35")
    t0, n0, l0 = timed_fit(lambda: build_blend(hsc, resizing=False)[0])
    t2, n2, l2 = timed_fit(lambda: build_blend(hsc, resizing=True)[0])
    blends = bench.build_facade_blends(0, 1, 0)
    import copy

    t1, n1, l1 = timed_fit(lambda: copy.deepcopy(blends[0]))
    print(json.dumps({
        "metric": "wall time of Blend.fit(100, e_rel=1e-4) on one scene (median of 5)",
        "configs[0] hsc_cosmos_35 quickstart blend (resizing off)": {
            "ms": round(t0 * 1e3, 2), "iterations": n0, "ms_per_iteration": round(t0 * 1e3 / n0, 4), "logL": l0},
        "configs[0] hsc_cosmos_35 quickstart blend (resizing on, the reference's default)": {
            "ms": round(t2 * 1e3, 2), "iterations": n2, "ms_per_iteration": round(t2 * 1e3 / n2, 4), "logL": l2},
        "configs[1] one synthetic 5x128x128 scene, 10 components 41x41 (resizing on)": {
            "ms": round(t1 * 1e3, 2), "iterations": n1, "ms_per_iteration": round(t1 * 1e3 / n1, 4), "logL": l1},
    }))


if __name__ == "__main__":
    main()
