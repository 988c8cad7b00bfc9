"""Known-answer tests for the two monotonic-mask operators of the reference's native
module, derived BY HAND from the C++ (scarlet/operators_pybind11.cc:61-232) on images
small enough to follow pixel by pixel -- the reference ships no test for them, and both
the oracle (oracle/csrc/mask.c) and the GPU kernels (csrc/mask.hip) are restatements.
Every case notes which lines of the C++ decide it.  Used by tests/test_oracle_golden.py
(oracle) and tests/test_gpu_parity.py (device)."""

import numpy as np

T, F = True, False

A = [[0.0, 0.1, 0.2, 0.1, 0.0],
     [0.1, 0.5, 0.6, 0.5, 0.1],
     [0.2, 0.6, 1.0, 0.9, 1.2],
     [0.1, 0.5, 0.6, 0.5, 0.1],
     [0.0, 0.1, 0.7, 0.1, 0.0]]


def _flags(shape, true_at):
    out = np.zeros(shape, dtype=bool)
    for ij in true_at:
        out[ij] = True
    return out


def valid_pixel_cases(dtype):
    """(name, image, start, variance, thresh, unchecked, orphans, bounds) after
    get_valid_monotonic_pixels from ``start`` with everything else unchecked."""
    corners = [(0, 0), (0, 4), (4, 0), (4, 4)]
    a = np.array(A, dtype=dtype)
    # V1: flood fill from the peak (2,2); a neighbour is accepted if it is lower than the
    # pixel it is reached from and above `thresh` (cc:75, 88, 101, 114).  Every positive
    # pixel has a higher accepted neighbour except (2,4) = 1.2 (its neighbours are 0.9,
    # 0.1, 0.1) and (4,2) = 0.7 (0.6, 0.1, 0.1); the corner zeros fail `> thresh`.  All six
    # are tested from an accepted neighbour, so they are orphans (cc:83, 96, 109, 122).
    left = corners + [(2, 4), (4, 2)]
    yield ("plain", a, (2, 2), 0.0, 0.0, _flags((5, 5), left), _flags((5, 5), left), [0, 4, 0, 4])
    # V2: variance 0.35 lets 1.2 < 0.9 + 0.35 and 0.7 < 0.6 + 0.35 pass; zeros still fail
    yield ("variance", a, (2, 2), 0.35, 0.0, _flags((5, 5), corners), _flags((5, 5), corners),
           [0, 4, 0, 4])
    # V3: the recursive calls do not pass `thresh` on (cc:81, 94, 107, 120 call with the
    # default 0), so it only applies to the four neighbours of the start pixel.  (1,2) =
    # 0.1 fails thresh = 0.15 when tested from the start, and is accepted later from
    # (1,1) = 0.5 with thresh 0.  (0,2) = 0.2 then has no higher accepted neighbour.
    b = a.copy()
    b[1, 2] = 0.1
    left = corners + [(2, 4), (4, 2), (0, 2)]
    yield ("thresh at the first level only", b, (2, 2), 0.0, 0.15, _flags((5, 5), left),
           _flags((5, 5), left), [0, 4, 0, 4])


def interpolation_cases(dtype):
    """(name, model, unchecked, orphans, rows, cols, recursive, bounds,
    model', unchecked', orphans', bounds') for linear_interpolate_invalid_pixels."""
    t = dtype
    corners = [(0, 0), (0, 4), (4, 0), (4, 4)]
    # I1: image A after V1, orphans in np.where order.  Per pixel (cc:166-196 gather the
    # linear extrapolations 2 m1 - m2 from the directions in which the profile rises,
    # cc:199 needs their sum > 0):
    #  (0,0) +i: 0.1 - (0.2 - 0.1), +j: the same -> sum 0: stays an orphan, set to 0 (cc:228)
    #  (0,4) +i would use (2,4), still unchecked -> "unchecked neighbours" (cc:168); -j gives
    #        0 -> no update, no longer unchecked, orphan flag kept (cc:224)
    #  (2,4) -i is skipped because of `i > 2` (cc:175); -j: 0.9 - (1.0 - 0.9) -> new value
    #  (4,0) -i: 0; +j: 0.1 - (0.7 - 0.1) < 0 [the comma operator at cc:184 only looks at
    #        (4,1), so the unchecked (4,2) does not block it] -> sum < 0: orphan, 0
    #  (4,2) -i: 0.6 - (1.0 - 0.6); -j skipped because of `j > 2` (cc:193) -> new value
    #  (4,4) uses the values (2,4) and (4,2) just got: sum < 0 -> orphan, 0
    a = np.array(A, dtype=t)
    want = a.copy()
    want[2, 4] = a[2, 3] - (a[2, 2] - a[2, 3])
    want[4, 2] = a[3, 2] - (a[2, 2] - a[3, 2])
    left = corners + [(2, 4), (4, 2)]
    oi, oj = np.where(_flags((5, 5), left))
    yield ("image A", a, _flags((5, 5), left), _flags((5, 5), left), oi, oj, True, [0, 4, 0, 4],
           want, _flags((5, 5), []), _flags((5, 5), corners), [0, 4, 0, 4])
    # I2: one row.  Target (0,2): -j is skipped by `j > 2` although (0,0) = 7 > (0,1) = 3
    # would add 3 - 4 = -1; +j: (0,4) = 5 > (0,3) = 3, and (0,4) being unchecked is not
    # seen (comma operator, cc:184) -> sum = 3 - 2 = 1 > 0.  (`j >= 2` or `||` give 0.)
    row = np.array([[7, 3, 9, 3, 5, 0, 0]], dtype=t)
    want = row.copy()
    want[0, 2] = 1
    yield ("row: j > 2 and the comma operator", row, _flags((1, 7), [(0, 2), (0, 4)]),
           _flags((1, 7), [(0, 2)]), np.array([0]), np.array([2]), True, [0, 0, 0, 6],
           want, _flags((1, 7), [(0, 4)]), _flags((1, 7), []), [0, 0, 0, 6])
    # I3: one column.  Target (2,0): -i skipped by `i > 2` (cc:175); +i uses (3,0), (4,0),
    # both checked (a proper || here, cc:168) -> 3 - 2 = 1.
    col = row.T.copy()
    want = col.copy()
    want[2, 0] = 1
    yield ("column: i > 2", col, _flags((7, 1), [(2, 0)]), _flags((7, 1), [(2, 0)]),
           np.array([2]), np.array([0]), True, [0, 6, 0, 0],
           want, _flags((7, 1), []), _flags((7, 1), []), [0, 6, 0, 0])
    # I4: as I3 but (4,0) is still unchecked: cc:168 sets "unchecked neighbours", nothing
    # is interpolated, the pixel leaves `unchecked` and keeps its value and orphan flag
    yield ("column: unchecked neighbour", col, _flags((7, 1), [(2, 0), (4, 0)]),
           _flags((7, 1), [(2, 0)]), np.array([2]), np.array([0]), True, [0, 6, 0, 0],
           col.copy(), _flags((7, 1), [(4, 0)]), _flags((7, 1), [(2, 0)]), [0, 6, 0, 0])


def sweep_transcription(img, weights, offsets, dist_idx, min_gradient):
    """operators_pybind11.cc:14-36 line by line in pure Python (scalar arithmetic in the
    image's dtype), independent of oracle/csrc/sweep.c:

        for d in range(dist_idx.size): didx = dist_idx[d]; ref_flux = 0
            for i in range(offsets.size):
                if weights(i, didx) > 0: ref_flux += flat_img(offsets[i] + didx) * weights(i, didx)
            flat_img(didx) = min(flat_img(didx), ref_flux * (1 - min_gradient))
    """
    t = img.dtype.type
    flat = img.reshape(-1)
    w = weights.astype(img.dtype)
    one_minus_g = t(1) - t(min_gradient)
    for d in range(len(dist_idx)):
        didx = int(dist_idx[d])
        ref_flux = t(0)
        for i in range(len(offsets)):
            if w[i, didx] > 0:
                ref_flux = t(ref_flux + t(flat[int(offsets[i]) + didx] * w[i, didx]))
        flat[didx] = min(flat[didx], t(ref_flux * one_minus_g))
    return img
