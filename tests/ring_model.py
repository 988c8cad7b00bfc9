"""Lane-level model of the ring schedule of the monotonic sweep (csrc/common.h: RingPlanHost,
csrc/kernels.hip: sweep_ring): the same loop in numpy, float32, separate multiply and add,
one row of 64 lanes per plane.  The plan comes from the library's host-side builder
(smi_sweep_ring_plan), so the test covers the builder and the schedule without a GPU."""
import ctypes

import numpy as np

from scarlet_amd import _lib


def ring_plan(shape, weights, offsets, didx):
    lib = _lib.load()
    h, w = shape
    wts64 = np.ascontiguousarray(weights, dtype=np.float64)
    off = _lib.i32(offsets)
    idx = _lib.i32(didx)
    info = np.zeros(8, dtype=np.int32)
    args = (h, w, _lib.ptr(wts64, ctypes.c_double), _lib.ptr(off, ctypes.c_int32),
            _lib.ptr(idx, ctypes.c_int32), idx.size, _lib.ptr(info, ctypes.c_int32))
    rc = _lib.check(lib.smi_sweep_ring_plan(*args, None, None, 0))
    if rc == 0:
        return None
    planes, n_steps, n_pad, rmax, centre, perm, lanes = (int(v) for v in info[:7])
    planes, n_nat = planes & 0xFF, planes >> 8
    wts = np.zeros((lanes, 4), dtype=np.float32)
    addr = np.zeros(lanes, dtype=np.uint16)
    _lib.check(lib.smi_sweep_ring_plan(*args, _lib.ptr(wts, ctypes.c_float),
                                       _lib.ptr(addr, ctypes.c_uint16), lanes))
    steps = lanes // (planes * 64)
    return dict(planes=planes, n_steps=n_steps, n_pad=n_pad, n_nat=n_nat, rmax=rmax, centre=centre,
                perm=perm & 0xFFFFFFFF, stream_bytes=int(info[7]), wts=wts.reshape(steps, planes, 64, 4),
                addr=addr.reshape(steps, planes, 64).astype(np.int64))


def run(plan, image, min_gradient):
    """The device loop on a flat float32 image; returns the swept copy."""
    f32 = np.float32
    P = plan["planes"]
    span = 8 * P
    # LDS image: byte address 16 + 4 pixel, address 0 = the spare cell
    lds = np.concatenate([np.zeros(4, dtype=f32), image.astype(f32)])
    lane = np.arange(64)
    m = lane & 7
    inner = m >= 1
    ror1 = (lane & ~15) | ((lane - 1) & 15)  # lane i reads lane i - 1 of its row of 16
    ror9 = (lane & ~15) | ((lane - 9) & 15)
    diag = np.where(lane & 8, (lane + 8) & 63, (lane - 8) & 63)
    code = (plan["perm"] >> (3 * (lane >> 3))) & 7
    asc, pd = (code & 1).astype(bool), code >> 1
    omg = f32(1) - f32(min_gradient)
    centre = lds[4 + plan["centre"]]
    out = np.full((P, 64), centre, dtype=f32)
    c1, c2, c3, c4 = out.copy(), out.copy(), out.copy(), out.copy()
    g2 = out.copy()  # the mirror lane's result of the step before the last
    for s in range(plan["n_pad"]):
        L = s + 1
        prev = out.copy()
        for p in range(P):
            below = prev[(p + P - 1) % P]  # plane that holds ring r - 1 of this plane's m = 0
            f_own = np.where(inner, prev[p][ror1], below[ror9])
            f_mir = np.where(inner, prev[p][ror9], below[ror1])
            c4[p], c3[p], c2[p], c1[p] = c3[p], c2[p], c1[p], f_own
            if s < plan["n_nat"]:
                # plain schedule: the level says where the axis and the diagonal pixels are
                A, B, C = c3[p].copy(), c2[p].copy(), c1[p]
                if L & 1:
                    ra = (L + 1) // 2
                    if ra <= plan["rmax"] and (ra % span) // 8 == p:
                        A = np.where(m == (ra & 7), f_mir, A)
                if (L + 1) % 3 == 0:
                    rd = (L + 1) // 3
                    if (rd % span) // 8 == p:
                        B = np.where(m == (rd & 7), prev[p][diag], B)
            else:
                # flagged part (one plane): late rings take operands one level older
                word = plan["addr"][s, p]
                late, axis, dg = (word & 0x8000) != 0, (word & 1) != 0, (word & 2) != 0
                A = np.where(axis, np.where(late, g2[p], f_mir), np.where(late, c4[p], c3[p]))
                B = np.where(dg, prev[p][diag], np.where(late, c3[p], c2[p]))
                C = np.where(late, c2[p], c1[p])
            g2[p] = f_mir
            w = plan["wts"][s, p]
            pA, pB, pC, pD = A * w[:, 0], B * w[:, 1], C * w[:, 2], prev[p] * w[:, 3]
            e0, e2 = np.where(asc, pA, pC), np.where(asc, pC, pA)
            q0 = np.where(pd == 0, pD, e0)
            q1 = np.where(pd == 0, e0, np.where(pd == 1, pD, pB))
            q2 = np.where(pd <= 1, pB, np.where(pd == 2, pD, e2))
            q3 = np.where(pd == 3, pD, e2)
            ref = f32(0) + q0
            ref = ref + q1
            ref = ref + q2
            ref = ref + q3
            lim = (ref * omg).astype(f32)
            a = plan["addr"][s, p]
            if s >= plan["n_nat"]:
                a = a & 0x7FFC  # (flags of the address word)
            a = a // 4
            cur = lds[a]
            new = np.where(lim < cur, lim, cur).astype(f32)
            lds[a] = new
            out[p] = new
    return lds[4:].copy()
