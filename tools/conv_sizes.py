"""Development aid: time of the fused convolution kernel per FFT shape (512 blends x 5 bands,
frames chosen to land on every supported length).  SCARLET_AMD_LIB selects another build."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401

from scarlet_amd import BlendBatch, ComponentSpec

rng = np.random.default_rng(3)
nb, C, p = 512, 5, 21
yy, xx = np.mgrid[:p, :p] - p // 2
kernel = np.exp(-(yy**2 + xx**2) / 8.0).astype(np.float32)[None]
kernel /= kernel.sum()
out = []
shapes = [(40, 40), (56, 56), (72, 72), (100, 100), (128, 128), (40, 100), (100, 40), (56, 128)]
if "--more" in sys.argv:
    shapes = [(128, 40), (128, 56), (128, 72), (72, 40), (40, 72), (56, 72), (72, 56), (72, 100), (100, 72)]
if "--wide" in sys.argv:
    shapes = [(100, 128), (72, 128), (40, 128), (128, 100), (56, 128), (128, 128)]
for H, W in shapes:
    data = rng.normal(0, 1, (nb, C, H, W)).astype(np.float32)
    weights = np.ones_like(data)
    morph = np.ones((11, 11), np.float32)
    comps = [[ComponentSpec(np.ones(C, np.float32), morph, (H // 2 - 5, W // 2 - 5))] for _ in range(nb)]
    b = BlendBatch(data, weights, comps, kernel=kernel, max_iter=40)
    b.set_sub_ranges(1)
    b.step(0, 5)
    b.enable_timing(True)
    b.step(5, 20)
    out.append("%dx%d->F%s %.4f" % (H, W, "x".join(map(str, b.fft_shape)), b.timing()["conv"]))
    b.close()
print(os.environ.get("SMI_CONV_WORKGROUP", "auto"), " ".join(out))
