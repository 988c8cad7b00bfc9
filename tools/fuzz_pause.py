"""Development aid: fit_blends with blends pausing at their own hooks against the lock-step
rounds (SCARLET_AMD_FIT_BLENDS=lockstep) for random iteration budgets, tolerances and warm
starts: iteration counts, losses, parameters, boxes bit for bit.

    python tools/fuzz_pause.py [n_cases] [seed]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import scarlet_amd as scarlet  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
bad = 0
for case in range(n_cases):
    n = int(rng.integers(1, 14))
    lo = int(rng.integers(0, 200))
    max_iter = int(rng.choice([1, 2, 5, 10, 11, 12, 20, 21, 22, 33, 47, 80]))
    min_iter = int(rng.choice([0, 1, 3, 15]))
    e_rel = float(rng.choice([1e-2, 1e-3, 1e-4]))
    second = int(rng.choice([0, 0, 7, 25]))  # a second call on the fitted blends (warm start)
    out = []
    for mode in ("", "lockstep"):
        if mode:
            os.environ["SCARLET_AMD_FIT_BLENDS"] = mode
        else:
            os.environ.pop("SCARLET_AMD_FIT_BLENDS", None)
        blends = bench.build_facade_blends(lo, lo + n, 0)
        res = scarlet.fit_blends(blends, max_iter, e_rel=e_rel, min_iter=min_iter)
        if second:
            res = res + scarlet.fit_blends(blends, second, e_rel=e_rel, min_iter=min_iter)
        out.append((res, blends))
    (ra, a), (rb, b) = out
    ok = ra == rb
    for x, y in zip(a, b):
        ok &= x.loss == y.loss
        for p, q in zip(x.parameters, y.parameters):
            ok &= p.shape == q.shape and np.array_equal(np.asarray(p), np.asarray(q))
            if ok and p.m is not None:
                ok &= np.array_equal(p.m, q.m) and np.array_equal(p.vhat, q.vhat)
    bad += not ok
    print("case %2d: n=%2d max_iter=%2d min_iter=%2d e_rel=%g second=%2d iterations %s  %s"
          % (case, n, max_iter, min_iter, e_rel, second, sorted({r[0] for r in ra}),
             "same" if ok else "DIFFERENT"), flush=True)
print("cases: %d; different: %d" % (n_cases, bad))
