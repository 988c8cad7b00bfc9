"""Development aid: shader clocks of the monotonic sweep alone (slot plan vs ring schedule) at
several occupancies, and a bit comparison of the two.
    python tools/sweep_cycles.py [side] [weighting]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401

from scarlet_amd import BlendBatch, ComponentSpec, synthetic, _lib

side = int(sys.argv[1]) if len(sys.argv) > 1 else 41
lib = _lib.load()
fn = lib.smi_debug_sweep_cycles
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_float,
               ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_longlong),
               ctypes.POINTER(ctypes.c_float)]
scenes = synthetic.make_batch(range(1234, 1236))
kern = synthetic.psfs()
morph = np.zeros((side, side), dtype=np.float32)
morph[side // 2, side // 2] = 1
comps = [[ComponentSpec(s["seds"][0], morph, (10, 10), sed_min_step=s["noise_rms"])] for s in scenes]
data = np.stack([s["data"] for s in scenes])
weights = np.stack([s["weights"] for s in scenes])
batch = BlendBatch(data, weights, comps, kernel=kern[2], max_iter=4)
n_rep = 20
images = {}
big = side > 47
ring_mode = int(os.environ.get("SWEEP_MODE", "2"))  # 2: stream staged in LDS, 3: read from L2
for waves, groups in (((1, 1), (1, 256), (4, 256), (6, 256), (6, 1024)) if big else
                      ((1, 1), (1, 256), (4, 256), (8, 256), (12, 256), (12, 1024))):
    row = []
    for mode in (0, ring_mode):
        nw = waves * groups
        cyc = np.zeros(nw, dtype=np.int64)
        img = np.zeros((nw, side * side), dtype=np.float32)
        _lib.check(fn(batch._h, 0, mode, n_rep, 0.0, waves, groups,
                      cyc.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)),
                      img.ctypes.data_as(ctypes.POINTER(ctypes.c_float))))
        images[mode] = img
        row.append((cyc.mean() / n_rep, cyc.max() / n_rep))
    same = np.array_equal(images[0].view(np.uint32), images[ring_mode].view(np.uint32))
    print("waves/group %2d groups %4d: slots %6.0f (max %6.0f)  ring, plan in LDS %6.0f (max %6.0f) clocks per sweep, same bits: %s"
          % (waves, groups, row[0][0], row[0][1], row[1][0], row[1][1], same), flush=True)
