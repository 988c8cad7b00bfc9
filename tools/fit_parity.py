"""Development aid: how closely the device follows the oracle over a whole fit of the
quickstart scene (hsc_cosmos_35, 100 iterations max, e_rel 1e-4), against the
reference-faithful oracle (float64 optimizer state) and against its float32-state mode."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import golden, hsc_scene
import scarlet_amd as amd

g = golden("hsc_cosmos_35")
for path in ("fused", "rocfft"):
    comps = [amd.ComponentSpec(g["sed_%d" % k], g["morph_%d" % k], g["origin_%d" % k],
                               sed_min_step=g["min_step_%d" % k]) for k in range(int(g["n_comp"]))]
    for precision in ("f32", "f64"):
        try:
            batch = amd.BlendBatch(g["images"][None], g["weights"][None], [comps],
                                   kernel=g["diff_kernel"], max_iter=100, conv_path=path,
                                   **({} if precision == "f32" else {"state_precision": "f64"}))
        except TypeError:
            continue
        n_iter, logL = batch.fit(max_iter=100, e_rel=1e-4)
        loss = batch.loss_history()[0]
        for name, kw in (("f64-state oracle", {}), ("f32-state oracle", dict(state_dtype=np.float32))):
            sc = hsc_scene(g, **kw)
            n_ref, _ = sc.fit(max_iter=100, e_rel=1e-4)
            m = min(len(loss), len(sc.loss))
            chi, ref = loss[:m] - sc.log_norm, np.array(sc.loss[:m]) - sc.log_norm
            rel = np.abs(chi - ref) / np.abs(ref)
            print("%-6s device %s vs %s: n_iter %d / %d, max rel chi2 diff %.2e (it %d), first 12 %.2e, "
                  "final %.2e" % (path, precision, name, n_iter[0], n_ref, rel.max(), rel.argmax(),
                                  rel[:12].max(), rel[m - 1]))
