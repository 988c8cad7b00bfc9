"""Proximal-operator factories with the reference's ``scarlet.operator`` names.

The monotonicity tables are built here on the host (once per box shape, like
the reference does in Python, operator.py:62-96, 591-667); the sweep itself runs
on the GPU through the C ABI (``smi_prox_weighted_monotonic_*`` for stand-alone
calls, the fused update kernel inside ``Blend.fit``).
"""

from functools import partial

import numpy as np

from . import _lib

# order of the 8 neighbours in every (8, N) table (reference operator.py:84)
NEIGHBOR_COORDS = [(-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1)]


def _center_or_default(shape, center):
    if center is None:
        return (shape[0] - 1) >> 1, (shape[1] - 1) >> 1
    return int(center[0]), int(center[1])


def sort_by_radius(shape, center=None):
    """Flat indices of an image of ``shape`` ordered by distance from ``center``."""
    cy, cx = _center_or_default(shape, center)
    yy = (np.arange(shape[0]) - cy)[:, None]
    xx = (np.arange(shape[1]) - cx)[None, :]
    return np.argsort(np.sqrt(xx**2 + yy**2).reshape(-1))


def getOffsets(width, coords=None):
    """Flat-index offsets of the neighbours and the slice pairs that align a
    vector with its shifted copy (reference operator.py:512-527)."""
    def aligned(offset):
        # v[fwd] lines up with v[back] when v is compared with itself ``offset`` apart
        if offset < 0:
            return slice(None, offset), slice(-offset, None)
        return slice(offset, None), slice(None, -offset)

    offsets = [dy * width + dx for dy, dx in (NEIGHBOR_COORDS if coords is None else coords)]
    pairs = [aligned(o) for o in offsets]
    return offsets, [fwd for fwd, _ in pairs], [back for _, back in pairs]


def diagonalizeArray(arr, shape=None, dtype=np.float64):
    """(8, N) table of every pixel's neighbour values and the mask of
    neighbours that do not exist (reference operator.py:530-572)."""
    if shape is None:
        height, width = arr.shape
        data = arr.reshape(-1)
    elif arr.ndim == 1:
        height, width = shape
        data = arr
    else:
        raise ValueError("Expected either a 2D array or a 1D array and a shape")
    n = height * width
    table = np.zeros((8, n), dtype=dtype)
    missing = np.ones((8, n), dtype=bool)
    yy, xx = np.divmod(np.arange(n), width)
    for i, (dy, dx) in enumerate(NEIGHBOR_COORDS):
        ok = (yy + dy >= 0) & (yy + dy < height) & (xx + dx >= 0) & (xx + dx < width)
        # the reference also fills the row-wrapped entries with data; they are masked
        off = dy * width + dx
        src = np.arange(n) + off
        inside = (src >= 0) & (src < n)
        table[i, inside] = data[src[inside]]
        missing[i] = ~ok
    return table, missing


def getRadialMonotonicWeights(shape, neighbor_weight="flat", center=None):
    """(8, N) float64 weights of the radial monotonicity operator.

    For pixel p and neighbour direction i the weight is non-zero only when that
    neighbour lies inside the image and strictly nearer the peak.  'angle'
    uses the cosine of the angle between the directions p->peak and
    p->neighbour, normalised to unit sum; 'flat' gives all such neighbours equal
    weight; 'nearest' keeps the single best aligned one
    (same tables as reference operator.py:591-667).
    """
    assert neighbor_weight in ["flat", "angle", "nearest"]
    h, w = shape
    if center is None:
        center = ((h - 1) // 2, (w - 1) // 2)
    py, px = int(center[0]), int(center[1])
    Y = (np.arange(h) - py)[:, None] * np.ones((1, w), dtype=int)
    X = (np.arange(w) - px)[None, :] * np.ones((h, 1), dtype=int)
    r2 = X * X + Y * Y
    flat = neighbor_weight == "flat"
    to_peak = None if flat else np.arctan2(-Y, -X)
    cosw = np.zeros((8, h, w))
    for i, (dy, dx) in enumerate(NEIGHBOR_COORDS):
        ny, nx = Y + dy, X + dx
        inside = (ny + py >= 0) & (ny + py < h) & (nx + px >= 0) & (nx + px < w)
        nearer = (nx * nx + ny * ny) < r2
        valid = inside & nearer
        if flat:
            # every neighbour that counts gets weight 1 below: its cosine (never exactly zero
            # for a strictly nearer neighbour -- pi / 2 is no float) need not be evaluated
            cosw[i][valid] = 1.0
        else:
            cosw[i][valid] = np.cos(to_peak[valid] - np.arctan2(float(dy), float(dx)))
    cosw = cosw.reshape(8, h * w)
    if neighbor_weight == "nearest":
        out = np.zeros_like(cosw)
        out[np.argmax(cosw, axis=0), np.arange(h * w)] = 1
        out[:, py * w + px] = 0
        return out
    if neighbor_weight == "flat":
        cosw[cosw != 0] = 1
    total = cosw.sum(axis=0)
    total[total == 0] = 1
    return cosw / total[None, :]


def _native_sweep(X, weights, offsets, didx, min_gradient):
    """``operators_pybind11.prox_weighted_monotonic`` through the C ABI, in place."""
    lib = _lib.load()
    if not X.flags.c_contiguous:
        raise ValueError("prox_weighted_monotonic needs a C-contiguous array")
    flat = X.reshape(-1)
    off = _lib.i32(offsets)
    idx = _lib.i32(didx)
    if flat.dtype == np.float32:
        wts = np.ascontiguousarray(weights, dtype=np.float32)
        fn, ct, mg = lib.smi_prox_weighted_monotonic_f32, _lib.ctypes.c_float, float(min_gradient)
    elif flat.dtype == np.float64:
        wts = np.ascontiguousarray(weights, dtype=np.float64)
        fn, ct, mg = lib.smi_prox_weighted_monotonic_f64, _lib.ctypes.c_double, float(min_gradient)
    else:
        raise TypeError("prox_weighted_monotonic: float32 or float64 array required")
    _lib.check(
        fn(_lib.ptr(flat, ct), _lib.ptr(wts, ct), _lib.ptr(off, _lib.ctypes.c_int32), off.size,
           _lib.ptr(idx, _lib.ctypes.c_int32), idx.size, flat.size, mg)
    )
    return X


_MANY_TABLES = {}


def prox_weighted_monotonic_many(images, centers, neighbor_weight="flat", min_gradient=0.1):
    """``prox_weighted_monotonic(shape, neighbor_weight, min_gradient, center=centers[i])``
    applied to ``images[i]`` for every i, in ONE launch through the C ABI
    (``smi_prox_weighted_monotonic_many_*``): the detection images of a scene's sources, each
    made monotonic about its own centre (source.py:312-333 inside the loop of
    initialization.py:287-363).  ``images`` (n, H, W), float32 or float64, C-contiguous, is
    modified in place and returned; every image comes out bit for bit as the single
    operator leaves it."""
    lib = _lib.load()
    if images.ndim != 3 or not images.flags.c_contiguous:
        raise ValueError("prox_weighted_monotonic_many needs a C-contiguous (n, H, W) array")
    n, h, w = images.shape
    if len(centers) != n:
        raise ValueError("one centre per image")
    if n == 0:
        return images
    if images.dtype == np.float32:
        fn, ct = lib.smi_prox_weighted_monotonic_many_f32, _lib.ctypes.c_float
    elif images.dtype == np.float64:
        fn, ct = lib.smi_prox_weighted_monotonic_many_f64, _lib.ctypes.c_double
    else:
        raise TypeError("prox_weighted_monotonic_many: float32 or float64 array required")
    shape = (h, w)
    weights = np.empty((n, 8, h * w), dtype=images.dtype)
    didx = np.empty((n, h * w - 1), dtype=np.int32)
    for i, center in enumerate(centers):
        # (windows about sources away from the border share one shape and one centre)
        key = (shape, neighbor_weight, (int(center[0]), int(center[1])), images.dtype.str)
        tables = _MANY_TABLES.get(key)
        if tables is None:
            tables = (getRadialMonotonicWeights(shape, neighbor_weight=neighbor_weight,
                                                center=center).astype(images.dtype),
                      sort_by_radius(shape, center)[1:].astype(np.int32))
            if len(_MANY_TABLES) >= 32:
                _MANY_TABLES.clear()
            _MANY_TABLES[key] = tables
        weights[i], didx[i] = tables
    off = _lib.i32(getOffsets(w)[0])
    _lib.check(fn(n, _lib.ptr(images, ct), h * w, _lib.ptr(weights, ct),
                  _lib.ptr(off, _lib.ctypes.c_int32), off.size,
                  _lib.ptr(didx, _lib.ctypes.c_int32), didx.shape[1], float(min_gradient)))
    return images


def _prox_weighted_monotonic(X, step, weights, didx, offsets, min_gradient=0.1):
    """Force a radially monotonic profile; ``X`` is modified in place."""
    return _native_sweep(X, weights, offsets, didx, min_gradient)


def prox_weighted_monotonic(shape, neighbor_weight="flat", min_gradient=0.1, center=None):
    """Build the monotonicity operator for images of ``shape``
    (reference operator.py:62-96).  Returns ``f(X, step) -> X``; the set-up tables are
    bound as keyword arguments, so ``f.keywords`` exposes them as the reference does."""
    order = sort_by_radius(shape, center)
    tables = dict(
        weights=getRadialMonotonicWeights(shape, neighbor_weight=neighbor_weight, center=center),
        didx=order[1:],  # the peak itself has no brighter neighbour to be bounded by
        offsets=np.array(getOffsets(shape[1])[0]),
        min_gradient=min_gradient,
    )
    return partial(_prox_weighted_monotonic, **tables)


_TABLES = {}


def monotonic_tables(shape, neighbor_weight, center=None):
    """(weights, offsets, didx without the peak) as int32/float64 arrays for the
    C ABI (``smi_batch_add_sweep_plan``); cached per (shape, weighting, centre) like
    the reference caches its operators (constraint.py:209-223)."""
    if center is None:
        center = (shape[0] // 2, shape[1] // 2)
    key = (tuple(shape), neighbor_weight, (int(center[0]), int(center[1])))
    if key not in _TABLES:
        _TABLES[key] = _monotonic_tables(shape, neighbor_weight, center)
    return _TABLES[key]


def _monotonic_tables(shape, neighbor_weight, center):
    weights = np.ascontiguousarray(
        getRadialMonotonicWeights(shape, neighbor_weight, center), dtype=np.float64
    )
    offsets = np.array([shape[1] * y + x for y, x in NEIGHBOR_COORDS], dtype=np.int32)
    didx = sort_by_radius(shape, center)[1:].astype(np.int32)
    return weights, offsets, didx


def get_center(image, center, radius=1):
    """Brightest pixel within ``radius`` of ``center`` (reference operator.py:99-129)."""
    cy, cx = int(center[0]), int(center[1])
    y0, x0 = max(cy - radius, 0), max(cx - radius, 0)
    patch = image[y0 : cy + radius + 1, x0 : cx + radius + 1]
    dy, dx = np.unravel_index(np.argmax(patch), patch.shape)
    return y0 + dy, x0 + dx


def _mask_args(image):
    if image.dtype == np.float32:
        return "f32", _lib.ctypes.c_float
    if image.dtype == np.float64:
        return "f64", _lib.ctypes.c_double
    raise TypeError("float32 or float64 array required")


def get_valid_monotonic_pixels(i, j, image, unchecked, orphans, variance, bounds, thresh=0):
    """``operators_pybind11.get_valid_monotonic_pixels`` through the C ABI: flood fill
    from ``(i, j)`` over pixels that do not rise by more than ``variance`` along the way;
    ``unchecked`` / ``orphans`` (bool) and ``bounds`` (int32[4]) are updated in place."""
    lib = _lib.load()
    tag, ct = _mask_args(image)
    if not (image.flags.c_contiguous and unchecked.flags.c_contiguous and orphans.flags.c_contiguous):
        raise ValueError("C-contiguous arrays required")
    assert unchecked.dtype == bool and orphans.dtype == bool and bounds.dtype == np.int32
    u8 = _lib.ctypes.c_uint8
    _lib.check(getattr(lib, "smi_get_valid_monotonic_pixels_" + tag)(
        int(i), int(j), _lib.ptr(image, ct), image.shape[0], image.shape[1],
        _lib.ptr(unchecked.view(np.uint8), u8), _lib.ptr(orphans.view(np.uint8), u8),
        float(variance), _lib.ptr(bounds, _lib.ctypes.c_int32), float(thresh)))


def linear_interpolate_invalid_pixels(row_indices, column_indices, unchecked, model, orphans,
                                      variance, recursive, bounds):
    """``operators_pybind11.linear_interpolate_invalid_pixels`` through the C ABI: fill the
    listed orphans from the gradients of their neighbours, in list order, in place."""
    lib = _lib.load()
    tag, ct = _mask_args(model)
    if not (model.flags.c_contiguous and unchecked.flags.c_contiguous and orphans.flags.c_contiguous):
        raise ValueError("C-contiguous arrays required")
    assert unchecked.dtype == bool and orphans.dtype == bool and bounds.dtype == np.int32
    rows, cols = _lib.i32(row_indices), _lib.i32(column_indices)
    u8, i32 = _lib.ctypes.c_uint8, _lib.ctypes.c_int32
    _lib.check(getattr(lib, "smi_linear_interpolate_invalid_pixels_" + tag)(
        _lib.ptr(rows, i32), _lib.ptr(cols, i32), rows.size,
        _lib.ptr(unchecked.view(np.uint8), u8), _lib.ptr(model, ct), model.shape[0], model.shape[1],
        _lib.ptr(orphans.view(np.uint8), u8), float(variance), int(bool(recursive)),
        _lib.ptr(bounds, i32)))


def prox_monotonic_mask(X, step, center, center_radius=1, variance=0.0, max_iter=3):
    """Monotonicity along *any* path from the centre (reference operator.py:131-176):
    pixels reachable from the (optionally re-fitted) centre without rising by more than
    ``variance`` are valid; orphans next to them are interpolated up to ``max_iter`` times;
    the rest is cleared.  Returns ``(valid, model, bounds)``."""
    seed = get_center(X, center, center_radius) if center_radius > 0 else np.round(center)
    i, j = (int(c) for c in seed)
    X = np.ascontiguousarray(X)

    # state the two native operators share: pixels not settled yet, pixels rejected from
    # every side so far, bounding box of the accepted region
    pending = np.ones(X.shape, dtype=bool)
    pending[i, j] = False
    rejected = np.zeros(X.shape, dtype=bool)
    box = np.array([i, i, j, j], dtype=np.int32)
    get_valid_monotonic_pixels(i, j, X, pending, rejected, variance, box, 0)
    model = X.copy()
    for _ in range(max_iter):
        if not np.any(rejected & pending):
            break
        rows, cols = np.nonzero(rejected)
        linear_interpolate_invalid_pixels(rows, cols, pending, model, rejected, variance, True, box)
    valid = ~(pending | rejected)
    return valid, np.where(valid, model, 0).astype(model.dtype), box


def prox_sdss_symmetry(X, step):
    """Minimum of each pixel and its 180-degree partner, in place."""
    X[:] = np.minimum(X, X[::-1, ::-1])
    return X


def prox_soft_symmetry(X, step, strength=1):
    """Blend ``X`` with its 180-degree rotation; even axes get one trailing
    zero before rotating (reference operator.py:274-293)."""
    h, w = X.shape
    padded = np.zeros((h + (h % 2 == 0), w + (w % 2 == 0)), dtype=X.dtype)
    padded[:h, :w] = X
    out = 0.5 * strength * (padded + padded[::-1, ::-1]) + (1 - strength) * padded
    return out[:h, :w]


def uncentered_operator(X, func, center=None, fill=None, **kwargs):
    """Apply ``func`` to the largest sub-array of ``X`` that is centred on
    ``center`` (reference operator.py:207-260)."""
    if center is None:
        py, px = np.unravel_index(np.argmax(X), X.shape)
    else:
        py, px = int(center[0]), int(center[1])
    cy, cx = np.array(X.shape) // 2
    if py == cy and px == cx:
        return func(X, **kwargs)
    dy, dx = int(2 * (py - cy)), int(2 * (px - cx))
    if not X.shape[0] % 2:
        dy += 1
    if not X.shape[1] % 2:
        dx += 1
    ysl = slice(dy, None) if dy > 0 else (slice(None, dy) if dy < 0 else slice(None))
    xsl = slice(dx, None) if dx > 0 else (slice(None, dx) if dx < 0 else slice(None))
    if fill is not None:
        out = np.ones(X.shape, X.dtype) * fill
        out[ysl, xsl] = func(X[ysl, xsl], **kwargs)
        X[:] = out
    else:
        X[ysl, xsl] = func(X[ysl, xsl], **kwargs)
    return X


def prox_uncentered_symmetry(X, step, center=None, algorithm="kspace", fill=None, shift=None,
                             strength=0.5):
    """Symmetry about an off-centre peak (reference operator.py:328-400); the
    k-space variant for fractional shifts is not supported."""
    if algorithm == "kspace":
        if shift is not None and not np.all(np.asarray(shift) == 0):
            raise NotImplementedError("k-space symmetry with a fractional shift")
        algorithm, strength = "soft", 1
    if algorithm == "sdss":
        return uncentered_operator(X, prox_sdss_symmetry, center, step=step, fill=fill)
    if algorithm == "soft":
        return uncentered_operator(
            X, prox_soft_symmetry, center, step=step, strength=strength, fill=fill
        )
    raise ValueError("algorithm must be 'sdss' or 'soft', got {}".format(algorithm))
