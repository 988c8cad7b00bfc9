"""Shared builder for the two-observation (high + low resolution) scene of
tests/golden/multires_fit.npz: the oracle's Scene with the low-resolution term."""
import numpy as np

from oracle import pgm, resample


def build(g, dtype=np.float64):
    """(scene, lowres observation, channel index of hr, slices of the hr data in the
    frame) from the golden's set-up quantities (the reference's own values)."""
    C, H, W = (int(v) for v in g["frame_shape"])
    channels = [str(c) for c in g["channels"]]
    c_hr, c_lr = channels.index("hr"), channels.index("lr")
    (dy0, dy1), (dx0, dx1), (my0, my1), (mx0, mx1) = g["hr_slices"]
    data = np.zeros((C, H, W))
    weights = np.zeros((C, H, W))
    data[c_hr, my0:my1, mx0:mx1] = g["data_hr"][0, dy0:dy1, dx0:dx1]
    weights[c_hr, my0:my1, mx0:mx1] = g["weights_hr"][0, dy0:dy1, dx0:dx1]
    stamp = g["hr_kernel"].shape[1:]
    kernel = np.zeros((C,) + stamp)
    kernel[c_hr] = g["hr_kernel"][0]
    kernel[c_lr, stamp[0] // 2, stamp[1] // 2] = 1  # unobserved there: weight zero
    lowres = resample.LowResObservation(
        g["lr_kernel"], g["lr_shifts"][0], g["lr_other_shifts"][1], float(g["lr_h"]), [c_lr],
        (H, W), g["data_lr"], g["weights_lr"])
    comps = [pgm.Component(g["sed_%d" % k].astype(np.float64), g["morph_%d" % k].copy(),
                           tuple(int(v) for v in g["origin_%d" % k]))
             for k in range(int(g["n_components"]))]
    scene = pgm.Scene((C, H, W), data.astype(dtype), weights.astype(dtype), kernel.astype(dtype),
                      comps, dtype=dtype, extra_observations=[lowres])
    return scene, lowres, c_hr, ((dy0, dy1), (dx0, dx1), (my0, my1), (mx0, mx1))
