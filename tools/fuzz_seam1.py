"""Seam 1 (the four functions of the reference's pybind11 module) on random inputs, GPU
against the oracle's C restatement, bit for bit: sweep (random shapes, centres,
weightings), apply_filter (random stamps), the two mask operators (random bumpy images,
start pixels, variance / threshold).  Development aid.

    python tools/fuzz_seam1.py [n_cases] [seed]
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from oracle import fftconv, proxops  # noqa: E402
from scarlet_amd import _lib, operator  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 9)
bad = []
lib = _lib.load()
olib = proxops._lib()
for n in range(n_cases):
    dtype = np.float32 if rng.random() < 0.5 else np.float64
    # ---- sweep
    shape = (int(rng.integers(2, 90)), int(rng.integers(2, 90)))  # the tables need >= 2 columns
    center = (int(rng.integers(0, shape[0])), int(rng.integers(0, shape[1])))
    mode = str(rng.choice(["angle", "flat", "nearest"]))
    g = float(rng.choice([0.0, 0.1, 0.5]))
    w, didx, off = proxops.monotonic_operator(shape, mode, center)
    x0 = rng.random(shape).astype(dtype)
    want = proxops.sweep(x0.copy(), w, off, didx, g)
    got = operator._native_sweep(x0.copy(), w, off, didx, g)
    if not np.array_equal(got, want):
        bad.append((n, "sweep", shape, center, mode, g, dtype.__name__))
    # ---- apply_filter
    H, W = int(rng.integers(1, 70)), int(rng.integers(1, 70))
    kh, kw = 2 * int(rng.integers(0, 6)) + 1, 2 * int(rng.integers(0, 6)) + 1
    img = rng.standard_normal((H, W)).astype(dtype)
    ker = rng.standard_normal((kh, kw)).astype(dtype)
    ys, ye, xs, xe = fftconv.filter_bounds(ker)
    vals = np.ascontiguousarray(ker.reshape(-1))
    want = np.empty_like(img)
    ofn = olib.oracle_apply_filter_f32 if dtype == np.float32 else olib.oracle_apply_filter_f64
    ofn.restype = None
    vp = ctypes.c_void_p
    ofn(img.ctypes.data_as(vp), H, W, vals.ctypes.data_as(vp), vals.size, ys.ctypes.data_as(vp),
        ye.ctypes.data_as(vp), xs.ctypes.data_as(vp), xe.ctypes.data_as(vp), want.ctypes.data_as(vp))
    got = np.empty_like(img)
    ct = ctypes.c_float if dtype == np.float32 else ctypes.c_double
    fn = lib.smi_apply_filter_f32 if dtype == np.float32 else lib.smi_apply_filter_f64
    _lib.check(fn(_lib.ptr(img, ct), H, W, _lib.ptr(vals, ct), vals.size,
                  _lib.ptr(ys, ctypes.c_int32), _lib.ptr(ye, ctypes.c_int32),
                  _lib.ptr(xs, ctypes.c_int32), _lib.ptr(xe, ctypes.c_int32), _lib.ptr(got, ct)))
    if not np.array_equal(got, want):
        bad.append((n, "apply_filter", (H, W), (kh, kw), dtype.__name__))
    # ---- mask operators
    shape = (int(rng.integers(1, 110)), int(rng.integers(2, 110)))
    yy, xx = np.mgrid[: shape[0], : shape[1]]
    cy, cx = rng.uniform(0, shape[0]), rng.uniform(0, shape[1])
    img = np.exp(-0.5 * (((yy - cy) / (0.25 * shape[0] + 1)) ** 2 + ((xx - cx) / (0.25 * shape[1] + 1)) ** 2))
    img += 0.5 * np.exp(-0.5 * (((yy - cy / 2) / 2.5) ** 2 + ((xx - 0.7 * cx) / 3.0) ** 2))
    img = (img + rng.normal(0, 0.03, shape)).astype(dtype)
    i, j = int(np.clip(round(cy), 0, shape[0] - 1)), int(np.clip(round(cx), 0, shape[1] - 1))
    variance = float(rng.choice([0.0, 0.01, 0.05]))
    thresh = float(rng.choice([0.0, 0.05]))
    state = []
    for mod in (proxops, operator):
        unchecked = np.ones(shape, dtype=bool)
        unchecked[i, j] = False
        orphans = np.zeros(shape, dtype=bool)
        bounds = np.array([i, i, j, j], dtype=np.int32)
        mod.get_valid_monotonic_pixels(i, j, img, unchecked, orphans, variance, bounds, thresh)
        model = img.copy()
        first = (unchecked.copy(), orphans.copy(), bounds.copy())
        for recursive in (True, False):
            oi, oj = np.where(orphans)
            mod.linear_interpolate_invalid_pixels(oi, oj, unchecked, model, orphans, variance,
                                                  recursive, bounds)
        state.append(first + (unchecked, orphans, bounds, model))
    for a, b in zip(*state):
        if not np.array_equal(a, b):
            bad.append((n, "mask", shape, (i, j), variance, thresh, dtype.__name__))
            break
print("seam-1 cases: %d x 3 operators; mismatches: %d" % (n_cases, len(bad)))
for entry in bad:
    print("OVER", entry)
