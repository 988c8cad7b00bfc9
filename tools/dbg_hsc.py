import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scarlet_amd as amd
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "hsc_cosmos_35.npz"))
n = int(g["n_comp"])
comps = [amd.ComponentSpec(g["sed_%d" % k], g["morph_%d" % k], g["origin_%d" % k], sed_min_step=g["min_step_%d" % k]) for k in range(n)]
b = amd.BlendBatch(g["images"][None], g["weights"][None], [comps], kernel=g["diff_kernel"], conv_path="fused")
model, rendered, logL = b.forward()
ref = g["rendered"]
d = np.abs(rendered[0] - ref)
print("fft", b.fft_shape, "max ref", np.abs(ref).max(), "max diff", d.max())
for c in range(d.shape[0]):
    print("band", c, "rows with diff > 1e-4 max:", np.where(d[c].max(axis=1) > 1e-4 * np.abs(ref).max())[0][:40], "cols:", np.where(d[c].max(axis=0) > 1e-4 * np.abs(ref).max())[0][:60])
