"""Prototype of the ring schedule of the monotonic sweep (host builder + a lane-level model).

The reference sweep (operators_pybind11.cc:14-36) clips every pixel, in order of radius,
against the weighted mean of its neighbours that are strictly nearer the peak.  For the
radial tables of operator.py:591-667 those neighbours are, with r = max(|Y|, |X|) and
j = min(|Y|, |X|) the pixel's ring and its position along the ring inside its octant,

    A = (r-1, j-1)   B = (r-1, j)   C = (r-1, j+1) [j < r-1]   D = (r, j-1),

so pixel (r, j) can be processed at level L = 2 r + j - 1 (A is final at L-3, B at L-2, C and
D at L-1).  One lane per (octant, r mod 8): the lane walks along its ring, j = 0 .. r, one pixel
per level, then waits for ring r + 8.  Its operands are its own last result (D) and the last
three results of the lane of ring r - 1 (A, B, C) -- no image reads besides the pixel itself.
Axis pixels (j = 0) take A from the mirror octant, diagonal pixels (j = r) take B from the
octant across the diagonal; both are computed in either octant.

This file is the executable specification: `build` mirrors csrc/sweep_plan.cpp's ring builder,
`run` the device loop of kernels.hip (float32 arithmetic, separate multiply and add).
"""
import numpy as np

NEIGHBOR_COORDS = [(-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1)]

# lane = row * 16 + half * 8 + (r mod 8); rows = pairs of octants sharing an axis
# (major axis, sign along it); halves = sign along the minor axis
ROWS = [("x", +1, (-1, +1)), ("y", +1, (+1, -1)), ("x", -1, (+1, -1)), ("y", -1, (-1, +1))]


def octant_roles(major, smaj, smin):
    """(dy, dx) of the roles A, B, C, D for a pixel of this octant."""
    if major == "x":
        sx, sy = smaj, smin
        return [(-sy, -sx), (0, -sx), (sy, -sx), (-sy, 0)]
    sy, sx = smaj, smin
    return [(-sy, -sx), (-sy, 0), (-sy, sx), (0, -sx)]


def build(shape, weights, offsets, didx):
    """Ring plan or None when the tables do not have the radial structure."""
    h, w = shape
    n = h * w
    order = np.full(n, -1)
    order[didx] = np.arange(len(didx))
    centre = np.flatnonzero(order < 0)
    if len(centre) != 1:
        return None
    cy, cx = divmod(int(centre[0]), w)
    if list(offsets) != [dy * w + dx for dy, dx in NEIGHBOR_COORDS]:
        return None
    rmax = max(cy, h - 1 - cy, cx, w - 1 - cx)
    if rmax > 23:
        return None
    n_steps = 3 * rmax - 1
    addr = np.full((n_steps + 1, 64), -1, dtype=np.int64)  # pixel index or -1 (idle)
    wts = np.zeros((n_steps + 1, 64, 4), dtype=np.float64)  # by role
    perm = np.zeros((64, 4), dtype=np.int64)  # roles in the order of the sum
    for row, (major, smaj, mins) in enumerate(ROWS):
        for half, smin in enumerate(mins):
            roles = octant_roles(major, smaj, smin)
            idx = [NEIGHBOR_COORDS.index(o) for o in roles]
            p = np.argsort(idx)
            for m in range(8):
                perm[row * 16 + half * 8 + m] = p
    seen = np.zeros(n, dtype=bool)
    for y in range(h):
        for x in range(w):
            Y, X = y - cy, x - cx
            if Y == 0 and X == 0:
                continue
            r, j = max(abs(Y), abs(X)), min(abs(Y), abs(X))
            L = 2 * r + j - 1
            pix = y * w + x
            slots = []
            for row, (major, smaj, mins) in enumerate(ROWS):
                maj_c, min_c = (X, Y) if major == "x" else (Y, X)
                if abs(maj_c) != r or np.sign(maj_c) != smaj or abs(min_c) != j:
                    continue
                for half, smin in enumerate(mins):
                    if min_c == 0 or np.sign(min_c) == smin:
                        slots.append((row, half, major, smaj, smin))
            assert slots
            for row, half, major, smaj, smin in slots:
                lane = row * 16 + half * 8 + (r & 7)
                roles = octant_roles(major, smaj, smin)
                idx = [NEIGHBOR_COORDS.index(o) for o in roles]
                used = set(np.flatnonzero(weights[:, pix] > 0))
                if not used <= set(idx):
                    return None
                for i in used:
                    q = pix + offsets[i]
                    if order[q] >= order[pix] and order[q] >= 0:
                        return None  # the sequential sweep would read a stale value
                assert addr[L, lane] < 0
                addr[L, lane] = pix
                wts[L, lane] = [weights[i, pix] for i in idx]
            seen[pix] = True
    if seen.sum() != len(didx):
        return None
    return dict(n_steps=n_steps, addr=addr, wts=wts.astype(np.float32), perm=perm,
                centre=int(centre[0]), rmax=rmax)


def lane_sources():
    lane = np.arange(64)
    m = lane & 7
    own = np.where(m >= 1, lane - 1, lane + 7)  # row_ror:1 / row_ror:9
    mir = np.where(m >= 1, (lane ^ 8) - 1, (lane ^ 8) + 7)  # row_ror:9 / row_ror:1
    diag = np.where(lane & 8, (lane + 8) & 63, (lane - 8) & 63)
    return own, mir, diag


def run(plan, image, min_gradient):
    """The device loop on a flat float32 image (modified in place)."""
    f32 = np.float32
    img = np.concatenate([image.astype(f32), [f32(0)]])  # [-1] = the spare cell
    own, mir, diag = lane_sources()
    omg = f32(1) - f32(min_gradient)
    centre = img[plan["centre"]]
    out = np.full(64, centre, dtype=f32)
    c1 = out.copy()
    c2 = out.copy()
    c3 = out.copy()
    lane = np.arange(64)
    m = lane & 7
    perm = plan["perm"]
    for L in range(1, plan["n_steps"] + 1):
        f_own, f_mir, f_diag = out[own], out[mir], out[diag]
        c3, c2, c1 = c2, c1, f_own
        axis = (L & 1) == 1 and (L + 1) // 2 <= plan["rmax"]  # ring r starts at level 2 r - 1
        ax_mask = (m == (((L + 1) // 2) & 7)) if axis else np.zeros(64, bool)
        dg_mask = (m == (((L + 1) // 3) & 7)) if (L + 1) % 3 == 0 else np.zeros(64, bool)
        A = np.where(ax_mask, f_mir, c3)
        B = np.where(dg_mask, f_diag, c2)
        vals = np.stack([A, B, c1, out], axis=1).astype(f32)
        prod = vals * plan["wts"][L]  # float32 products
        s = np.zeros(64, dtype=f32)
        for k in range(4):
            s = s + prod[lane, perm[:, k]]
        lim = s * omg
        a = plan["addr"][L]
        cur = img[a]
        new = np.where(lim < cur, lim, cur).astype(f32)
        img[a] = new
        img[-1] = 0
        out = new
    image[:] = img[:-1]
    return image


def sequential(image, weights, offsets, didx, min_gradient):
    f32 = np.float32
    w32 = weights.astype(f32)
    omg = f32(1) - f32(min_gradient)
    for p in didx:
        ref = f32(0)
        for i in range(len(offsets)):
            if weights[i, p] > 0:
                ref = f32(ref + f32(image[p + offsets[i]] * w32[i, p]))
        lim = f32(ref * omg)
        if lim < image[p]:
            image[p] = lim
    return image


if __name__ == "__main__":
    import sys
    sys.path.insert(0, ".")
    from scarlet_amd.operator import getRadialMonotonicWeights, sort_by_radius

    rng = np.random.default_rng(5)
    shapes = [(5, 5), (3, 3), (21, 21), (31, 31), (41, 41), (45, 45), (47, 47), (31, 41), (22, 30),
              (40, 40), (41, 40), (7, 47), (46, 11)]
    for shape in shapes:
        for kind in ["angle", "flat", "nearest"]:
            for centre in [None, "shift"]:
                c = (shape[0] // 2, shape[1] // 2)
                if centre == "shift":
                    c = (min(c[0] + 1, shape[0] - 1), max(c[1] - 1, 0))
                wt = getRadialMonotonicWeights(shape, kind, c)
                offsets = np.array([shape[1] * y + x for y, x in NEIGHBOR_COORDS])
                didx = sort_by_radius(shape, c)[1:]
                plan = build(shape, wt, offsets, didx)
                if plan is None:
                    print(shape, kind, centre, "no ring plan")
                    continue
                for g in [0.0, 0.25]:
                    img = rng.random(shape[0] * shape[1]).astype(np.float32)
                    ref = sequential(img.copy(), wt, offsets, didx, g)
                    got = run(plan, img.copy(), g)
                    assert np.array_equal(ref.view(np.uint32), got.view(np.uint32)), (shape, kind, g)
                active = (plan["addr"][1:] >= 0).sum()
                print(shape, kind, centre, "ok: steps", plan["n_steps"], "lane-steps", active,
                      "of", plan["n_steps"] * 64)
