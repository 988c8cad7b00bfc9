import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scarlet_amd import BlendBatch, ComponentSpec, synthetic, _lib
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
kern = synthetic.psfs()
scenes = synthetic.make_batch(range(1234, 1234 + nb), kernel=kern)
comps = [[ComponentSpec(s["seds"][k], s["morphs"][k], s["origins"][k], sed_min_step=s["noise_rms"]) for k in range(10)] for s in scenes]
b = BlendBatch(np.stack([s["data"] for s in scenes]), np.stack([s["weights"] for s in scenes]), comps, kernel=kern[2], max_iter=64)
b.set_sub_ranges(1)
lib = _lib.load()
out = (ctypes.c_longlong * 16)()
lib.smi_debug_fused_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
lib.smi_debug_fused_stamps(b._h, out)
b.step(0, 5)
b.status()
lib.smi_debug_fused_stamps(b._h, out)
t = list(out)
names = ["A rows fwd+render", "B columns", "C inv+resid+fwd", "B' columns", "D rows inv+store"]
for i, n in enumerate(names):
    print("%-22s %8d cycles" % (n, t[i + 1] - t[i]))
print("total", t[5] - t[0])
print("stage A: stride pass (loads -> T)", t[6]-t[0], "radix-16 + separation", t[1]-t[6])
print("stage C: radix-16 inverse", t[10]-t[2], "inverse + residual + forward stride pass", t[11]-t[10], "radix-16 + separation + loss", t[3]-t[11])
print("stage D: radix-16 inverse", t[12]-t[4], "stride pass + stores", t[5]-t[12])
