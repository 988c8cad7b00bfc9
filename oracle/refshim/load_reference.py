"""Import the read-only reference checkout under container-only shims.

BUILD-CONTAINER TOOLING.  It does nothing useful on the GPU box (the reference
is not there) and is never imported by the product, the tests, smoke() or
bench.py.  Its only client is ``make_golden.py`` next to it, which writes the
golden vectors committed under ``tests/golden/``.

Shims (all ours, under ``shims/``): ``autograd`` (NumPy passthrough, no AD),
``proxmin`` (operators from ``oracle.proxops``; no ``adaprox``), ``astropy``
(empty WCS class), and in-memory stand-ins for the reference's two pybind11
extensions (Eigen is absent, they cannot be compiled here) that forward to the
oracle's C restatement.
"""

import ctypes
import os
import sys
import types

import numpy as np

REFERENCE = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))


def _native_stub():
    from oracle import proxops

    mod = types.ModuleType("scarlet.operators_pybind11")

    def prox_weighted_monotonic(flat_img, weights, offsets, dist_idx, min_gradient):
        proxops.sweep(flat_img, weights, offsets, dist_idx, min_gradient)

    def apply_filter(image, values, y_start, y_end, x_start, x_end, result):
        lib = proxops._lib()
        if image.dtype == np.float32:
            fn = lib.oracle_apply_filter_f32
        else:
            fn = lib.oracle_apply_filter_f64
        fn.restype = None
        img = np.ascontiguousarray(image)
        vals = np.ascontiguousarray(values, dtype=image.dtype)
        args = [np.ascontiguousarray(a, dtype=np.int32) for a in (y_start, y_end, x_start, x_end)]
        out = np.empty(img.shape, dtype=img.dtype)
        fn(
            img.ctypes.data_as(ctypes.c_void_p),
            ctypes.c_int(img.shape[0]),
            ctypes.c_int(img.shape[1]),
            vals.ctypes.data_as(ctypes.c_void_p),
            ctypes.c_int(vals.size),
            *[a.ctypes.data_as(ctypes.c_void_p) for a in args],
            out.ctypes.data_as(ctypes.c_void_p),
        )
        result[...] = out

    mod.prox_weighted_monotonic = prox_weighted_monotonic
    mod.apply_filter = apply_filter
    # the mask operators forward to the oracle's C restatement as well, so that the
    # reference's own Python around them (operator.prox_monotonic_mask, the lite
    # initialisation with use_mask=True) can run
    mod.get_valid_monotonic_pixels = proxops.get_valid_monotonic_pixels
    mod.linear_interpolate_invalid_pixels = proxops.linear_interpolate_invalid_pixels
    return mod


def load():
    """Return the imported reference package ``scarlet``."""
    if not os.path.isdir(os.path.join(REFERENCE, "scarlet")):
        raise RuntimeError("reference checkout not present; golden vectors are committed")
    sys.dont_write_bytecode = True
    for p in (REFERENCE, REPO, os.path.join(HERE, "shims")):
        if p not in sys.path:
            sys.path.insert(0, p)
    sys.modules.setdefault("scarlet.operators_pybind11", _native_stub())
    detect = types.ModuleType("scarlet.detect_pybind11")
    for name in ("get_connected_pixels", "get_peaks", "get_footprints", "Footprint", "Peak"):
        setattr(detect, name, None)
    sys.modules.setdefault("scarlet.detect_pybind11", detect)
    import scarlet  # noqa: E402

    return scarlet
