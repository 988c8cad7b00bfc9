"""Oracle: proximal operators on the PGM path (``scarlet/constraint.py``,
``scarlet/operator.py``, ``scarlet/operators_pybind11.cc`` and the
``proxmin.operators`` functions the reference calls).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

``proxmin`` (>=0.6.11, reference setup.py:140) is not vendored and not
installed; ``prox_plus / prox_soft / prox_hard / prox_hard_plus /
prox_unity_plus`` restate its published definitions and are pinned through the
reference's call sites and tests (tests/test_constraint.py:35-90).
"""

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# neighbour order used throughout the reference (operator.py:84, 523)
NEIGHBOURS = ((-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1))


def _lib():
    """Load (building on first use when a compiler is present) liboracle.so."""
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            import subprocess

            subprocess.check_call(["make", "-C", _HERE, "liboracle.so"])
        _LIB = ctypes.CDLL(path)
    return _LIB


# --------------------------------------------------------------------------
# monotonic operator set-up (operator.py:10-48, 62-96, 512-667)
# --------------------------------------------------------------------------


def sort_by_radius(shape, center=None):
    """Flat pixel indices by increasing Euclidean distance from ``center``
    (operator.py:10-48).  Default centre ((h-1)>>1, (w-1)>>1)."""
    h, w = shape
    if center is None:
        cy, cx = (h - 1) >> 1, (w - 1) >> 1
    else:
        cy, cx = int(center[0]), int(center[1])
    yy, xx = np.meshgrid(np.arange(h) - cy, np.arange(w) - cx, indexing="ij")
    return np.argsort(np.sqrt(xx**2 + yy**2).flatten())


def _diagonalize(arr2d):
    """``diagonalizeArray`` (operator.py:530-572): (8, N) table of each pixel's
    8 neighbours' values plus the mask of neighbours that do not exist
    (outside the flat vector, or wrapped around a row end)."""
    h, w = arr2d.shape
    n = h * w
    flat = arr2d.reshape(-1)
    diag = np.zeros((8, n), dtype=np.float64)
    mask = np.ones((8, n), dtype=bool)
    for i, (dy, dx) in enumerate(NEIGHBOURS):
        off = dy * w + dx
        if off < 0:
            diag[i, -off:] = flat[:off]
            mask[i, -off:] = False
        else:
            diag[i, :-off] = flat[off:]
            mask[i, :-off] = False
    # wrapped "neighbours" of edge pixels (operator.py:561-570)
    mask[0][np.arange(1, h) * w] = True
    mask[2][np.arange(h) * w - 1] = True
    mask[3][np.arange(1, h) * w] = True
    mask[4][np.arange(1, h) * w - 1] = True
    mask[5][np.arange(h) * w] = True
    mask[7][np.arange(1, h - 1) * w - 1] = True
    return diag, mask


def radial_monotonic_weights(shape, neighbor_weight="flat", center=None):
    """``getRadialMonotonicWeights`` (operator.py:591-667): (8, N) float64.

    Row i is the weight of neighbour ``NEIGHBOURS[i]``; only neighbours
    strictly nearer the peak get weight ('angle': cosine of the angle between
    the direction to the peak and the direction to the neighbour, normalised;
    'flat': equal; 'nearest': one-hot on the best aligned one).
    """
    assert neighbor_weight in ("flat", "angle", "nearest")
    h, w = shape
    if center is None:
        center = ((h - 1) // 2, (w - 1) // 2)
    py, px = int(center[0]), int(center[1])
    Y, X = np.meshgrid(np.arange(h) - py, np.arange(w) - px, indexing="ij")
    dist = np.sqrt(X**2 + Y**2)

    dist_nb, mask = _diagonalize(dist)
    invalid = (dist.reshape(-1)[None, :] - dist_nb) <= 0

    # angle of the pixel -> peak direction (operator.py:616-622)
    on_axis = X == 0
    tX = X.copy()
    tX[on_axis] = 1
    ang = np.arctan2(-Y, -tX)
    sel = on_axis & (Y != 0)
    ang[sel] = 0.5 * np.pi * np.sign(ang[sel])

    # angle of the pixel -> neighbour direction (operator.py:624-635)
    x_nb, _ = _diagonalize(X)
    y_nb, _ = _diagonalize(Y)
    dx = x_nb - X.reshape(-1)[None, :]
    dy = y_nb - Y.reshape(-1)[None, :]
    vert = dx == 0
    dx[vert] = 1
    rel = np.arctan2(dy, dx)
    sel = vert & (dy != 0)
    rel[sel] = 0.5 * np.pi * np.sign(rel[sel])

    cosw = np.cos(ang.reshape(-1)[None, :] - rel)
    cosw[invalid] = 0
    cosw[mask] = 0

    if neighbor_weight == "nearest":
        out = np.zeros_like(cosw)
        out[np.argmax(cosw, axis=0), np.arange(cosw.shape[1])] = 1
        out[:, px + py * w] = 0
        return out
    if neighbor_weight == "flat":
        cosw[cosw != 0] = 1
    norm = cosw.sum(axis=0)
    norm[norm == 0] = 1
    out = cosw / norm[None, :]
    out[mask] = 0
    return out


def monotonic_operator(shape, neighbor_weight="flat", center=None):
    """The pieces ``operator.prox_weighted_monotonic`` binds (operator.py:62-96):
    weights (8,N) f64, sweep order without the peak (``didx[1:]``), offsets."""
    h, w = shape
    didx = sort_by_radius(shape, center)
    offsets = np.array([w * dy + dx for dy, dx in NEIGHBOURS], dtype=np.int32)
    weights = radial_monotonic_weights(shape, neighbor_weight, center)
    return weights, didx[1:].astype(np.int32), offsets


def sweep(img, weights, offsets, dist_idx, min_gradient):
    """In-place sequential sweep (operators_pybind11.cc:14-36) on ``img``
    (2-D or flat, float32 or float64, C-contiguous).  As in the pybind11
    overload set, the weights are converted to the image's dtype."""
    assert img.flags.c_contiguous, "sweep works in place on a C-contiguous image"
    flat = img.reshape(-1)
    wts = np.ascontiguousarray(weights, dtype=flat.dtype)
    off = np.ascontiguousarray(offsets, dtype=np.int32)
    idx = np.ascontiguousarray(dist_idx, dtype=np.int32)
    if flat.dtype == np.float32:
        fn, ct = _lib().oracle_prox_weighted_monotonic_f32, ctypes.c_float
    elif flat.dtype == np.float64:
        fn, ct = _lib().oracle_prox_weighted_monotonic_f64, ctypes.c_double
    else:
        raise TypeError("sweep: float32 or float64 image required")
    fn.restype = None
    fn(
        flat.ctypes.data_as(ctypes.c_void_p),
        wts.ctypes.data_as(ctypes.c_void_p),
        off.ctypes.data_as(ctypes.c_void_p),
        ctypes.c_int(off.size),
        idx.ctypes.data_as(ctypes.c_void_p),
        ctypes.c_int(idx.size),
        ctypes.c_int(flat.size),
        ct(min_gradient),
    )
    return img


def sweep_py(img, weights, offsets, dist_idx, min_gradient):
    """Pure-Python statement of the same loop (small cases only)."""
    flat = img.reshape(-1)
    for p in dist_idx:
        ref = flat.dtype.type(0)
        for i in range(len(offsets)):
            wi = flat.dtype.type(weights[i, p])
            if wi > 0:
                ref = ref + flat[p + offsets[i]] * wi
        lim = ref * (flat.dtype.type(1) - flat.dtype.type(min_gradient))
        if lim < flat[p]:
            flat[p] = lim
    return img


_OPS = {}


def prox_monotonic(morph, step=0, neighbor_weight="flat", min_gradient=0.1):
    """``MonotonicityConstraint.__call__`` with ``use_mask=False`` and a fixed
    centre (constraint.py:203-234): centre (h//2, w//2), operator cached per
    (shape, centre, weighting); the sweep mutates ``morph`` and returns it."""
    shape = morph.shape
    center = (shape[0] // 2, shape[1] // 2)
    key = (shape, center, neighbor_weight)
    if key not in _OPS:
        _OPS[key] = monotonic_operator(shape, neighbor_weight, center)
    weights, didx, offsets = _OPS[key]
    return sweep(morph, weights, offsets, didx, min_gradient)


# --------------------------------------------------------------------------
# element-wise constraints (constraint.py:83-114, 262-287)
# --------------------------------------------------------------------------


def prox_positivity(x, step=0, zero=0):
    """``PositivityConstraint`` (constraint.py:83-92)."""
    return np.maximum(x, zero)


def prox_center_on(morph, step=0, tiny=1e-6):
    """``CenterOnConstraint`` (constraint.py:276-287), in place."""
    c = (morph.shape[0] // 2, morph.shape[1] // 2)
    morph[c] = max(morph[c], tiny)
    return morph


def prox_normalization(x, step=0, type="max"):
    """``NormalizationConstraint`` (constraint.py:95-114), in place."""
    if type == "sum":
        x /= x.sum()
    else:
        x /= x.max()
    return x


def prox_soft_symmetry(x, step=0, strength=1):
    """``prox_soft_symmetry`` (operator.py:274-293): blend with the 180-degree
    rotation; even-sized axes are padded by one trailing zero first."""
    h, w = x.shape
    ph, pw = int(h % 2 == 0), int(w % 2 == 0)
    xp = np.zeros((h + ph, w + pw), dtype=x.dtype)
    xp[:h, :w] = x
    xs = xp[::-1, ::-1]
    out = 0.5 * strength * (xp + xs) + (1 - strength) * xp
    return out[:h, :w]


# --------------------------------------------------------------------------
# proxmin.operators as called by the reference (third party, restated)
# --------------------------------------------------------------------------


def _thresh(step, thresh, type):
    assert type in ("relative", "absolute")
    return thresh * step if type == "relative" else thresh


def prox_plus(x, step=0):
    """Projection on the non-negative orthant (used via constraint.py:163,
    operator.py:4,508)."""
    x[x < 0] = 0
    return x


def prox_soft(x, step, thresh=0, type="relative"):
    """Soft thresholding, ``sign(x) max(|x|-t, 0)`` (constraint.py:134-145;
    semantics pinned by tests/test_constraint.py:54-71)."""
    t = _thresh(step, thresh, type)
    return np.sign(x) * prox_plus(np.abs(x) - t)


def prox_hard(x, step, thresh=0, type="relative"):
    """Hard thresholding, ``|x| < t -> 0`` (constraint.py:117-130;
    tests/test_constraint.py:35-52), in place."""
    t = _thresh(step, thresh, type)
    x[np.abs(x) < t] = 0
    return x


def prox_hard_plus(x, step, thresh=0, type="relative"):
    """Hard thresholding on the non-negative orthant, ``x < t -> 0``
    (constraint.py:163; tests/test_constraint.py:73-90), in place."""
    t = _thresh(step, thresh, type)
    x[x < t] = 0
    return x


def prox_unity_plus(x, step=0, axis=0):
    """Non-negative, unit sum along ``axis`` (operator.py:4,508)."""
    x = prox_plus(x)
    return x / np.sum(x, axis=axis, keepdims=True)


def threshold_value(morph):
    """``ThresholdConstraint.threshold`` (constraint.py:165-180): last empty
    bin of the log10 histogram of the positive pixels."""
    pos = morph[morph > 0]
    bins = 50
    if pos.size < 500:
        bins = max(int(pos.size / 10), 1)
        if bins == 1:
            return 0, bins
    hist, edges = np.histogram(np.log10(pos).reshape(-1), bins)
    empty = np.where(hist == 0)[0]
    if len(empty) == 0:
        return 0, bins
    return 10 ** edges[empty[-1]], bins


def prox_threshold(x, step=0):
    """``ThresholdConstraint.__call__`` (constraint.py:161-163)."""
    t, _ = threshold_value(x)
    return prox_hard_plus(x, step, thresh=t, type="absolute")


def morph_chain(morph, step=0, monotonic="angle", min_gradient=0.0, symmetric=False,
                sparsity=None, tiny=1e-6, repeat=1, zero=0):
    """The ``ExtendedSourceMorphology`` constraint chain (morphology.py:644-670):
    Monotonicity -> [Symmetry] -> Positivity -> CenterOn -> Normalization("max"),
    optionally with an ``L0Constraint`` / ``L1Constraint`` (``sparsity`` =
    ("l0" | "l1", thresh, type), constraint.py:117-145) after the symmetry, the order a
    user chain must have to run on the device."""
    x = morph
    if repeat > 1:  # ConstraintChain(repeat) (constraint.py:60-80): the whole chain again
        for _ in range(repeat):
            x = morph_chain(x, step, monotonic, min_gradient, symmetric, sparsity, tiny, 1, zero)
        return x
    if monotonic is not None:
        x = prox_monotonic(x, step, monotonic, min_gradient)
    if symmetric:
        # True = SymmetryConstraint() (strength 1); a number = SymmetryConstraint(strength)
        x = prox_soft_symmetry(x, step, 1 if symmetric is True else symmetric)
    if sparsity is not None:
        kind, thresh, type = sparsity
        x = (prox_hard if kind == "l0" else prox_soft)(x, step, thresh, type)
    x = prox_positivity(x, step, zero)  # PositivityConstraint(zero), constraint.py:83-92
    x = prox_center_on(x, step, tiny)
    x = prox_normalization(x, step, "max")
    return x


# ---------------------------------------------------------------------------
# monotonic mask operators (operators_pybind11.cc:61-232, operator.py:131-176)
# ---------------------------------------------------------------------------
def _as_bytes(a):
    assert a.dtype == bool and a.flags.c_contiguous
    return a.view(np.uint8)


def get_valid_monotonic_pixels(i, j, image, unchecked, orphans, variance, bounds, thresh=0):
    """In place on ``unchecked`` / ``orphans`` (bool) and ``bounds`` (int32[4])."""
    assert image.flags.c_contiguous and bounds.dtype == np.int32
    name = "oracle_get_valid_monotonic_pixels_" + ("f32" if image.dtype == np.float32 else "f64")
    fn = getattr(_lib(), name)
    fn.restype = None
    vp = ctypes.c_void_p
    fn(ctypes.c_int(int(i)), ctypes.c_int(int(j)), image.ctypes.data_as(vp),
       ctypes.c_int(image.shape[0]), ctypes.c_int(image.shape[1]),
       _as_bytes(unchecked).ctypes.data_as(vp), _as_bytes(orphans).ctypes.data_as(vp),
       ctypes.c_double(variance), bounds.ctypes.data_as(vp), ctypes.c_double(thresh))


def linear_interpolate_invalid_pixels(row_indices, column_indices, unchecked, model, orphans,
                                      variance, recursive, bounds):
    assert model.flags.c_contiguous and bounds.dtype == np.int32
    rows = np.ascontiguousarray(row_indices, dtype=np.int32)
    cols = np.ascontiguousarray(column_indices, dtype=np.int32)
    name = "oracle_linear_interpolate_invalid_pixels_" + (
        "f32" if model.dtype == np.float32 else "f64")
    fn = getattr(_lib(), name)
    fn.restype = None
    vp = ctypes.c_void_p
    fn(rows.ctypes.data_as(vp), cols.ctypes.data_as(vp), ctypes.c_int(rows.size),
       _as_bytes(unchecked).ctypes.data_as(vp), model.ctypes.data_as(vp),
       ctypes.c_int(model.shape[0]), ctypes.c_int(model.shape[1]),
       _as_bytes(orphans).ctypes.data_as(vp), ctypes.c_double(variance),
       ctypes.c_int(int(bool(recursive))), bounds.ctypes.data_as(vp))


def prox_monotonic_mask(X, step, center, center_radius=1, variance=0.0, max_iter=3):
    """``operator.prox_monotonic_mask`` (operator.py:131-176): pixels reachable from the
    centre along a non-increasing path are valid; orphans are interpolated from the
    gradients of their neighbours up to ``max_iter`` times; everything else is cleared.
    Returns ``(valid, model, bounds)``."""
    if center_radius > 0:
        cy, cx = int(center[0]), int(center[1])
        y0, x0 = max(cy - center_radius, 0), max(cx - center_radius, 0)
        sub = X[y0 : cy + center_radius + 1, x0 : cx + center_radius + 1]
        di, dj = np.unravel_index(np.argmax(sub), sub.shape)
        i, j = di + y0, dj + x0
    else:
        i, j = int(np.round(center[0])), int(np.round(center[1]))
    unchecked = np.ones(X.shape, dtype=bool)
    unchecked[i, j] = False
    orphans = np.zeros(X.shape, dtype=bool)
    bounds = np.array([i, i, j, j], dtype=np.int32)
    get_valid_monotonic_pixels(i, j, X, unchecked, orphans, variance, bounds, 0)
    model = X.copy()
    it = 0
    while np.sum(orphans & unchecked) > 0 and it < max_iter:
        it += 1
        all_i, all_j = np.where(orphans)
        linear_interpolate_invalid_pixels(all_i, all_j, unchecked, model, orphans, variance, True,
                                          bounds)
    valid = ~unchecked & ~orphans
    return valid, model * valid, bounds
