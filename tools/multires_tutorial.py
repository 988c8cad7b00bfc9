"""The reference's multi-resolution tutorial (docs/tutorials/multiresolution.ipynb) through
the facade, from the committed fixture: timings of set-up, initialisation and fit.

    python tools/multires_tutorial.py [iterations]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import scarlet_amd as scarlet  # noqa: E402

g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden",
                         "multires_tutorial.npz"))
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 50


def wcs(tag, n):
    return scarlet.TanWCS(g["crpix_" + tag], g["crval_" + tag], g["pc_" + tag], g["cdelt_" + tag],
                          array_shape=(n, n))


t0 = time.time()
obs_hst = scarlet.Observation(g["data_hst"].copy(), wcs=wcs("hst", 250),
                              psf=scarlet.ImagePSF(g["psf_hst"].copy()),
                              channels=[str(c) for c in g["channels_hst"]])
obs_hsc = scarlet.Observation(g["data_hsc"].copy(), wcs=wcs("hsc", 50),
                              psf=scarlet.ImagePSF(g["psf_hsc"].copy()),
                              channels=[str(c) for c in g["channels_hsc"]])
observations = [obs_hsc, obs_hst]
frame = scarlet.Frame.from_observations(observations, coverage="intersection",
                                        model_psf=scarlet.GaussianPSF(sigma=0.6))
t1 = time.time()
ra_dec = obs_hst.get_sky_coord(g["pixel_hst"])
sources = [scarlet.ExtendedSource(frame, sky, observations, thresh=0.1) for sky in ra_dec]
scarlet.initialization.set_spectra_to_match(sources, observations)
blend = scarlet.Blend(sources, observations)
t2 = time.time()
n, logL = blend.fit(n_iter, e_rel=1e-4)
t3 = time.time()
print("frame %s, %d sources" % (tuple(frame.shape), len(sources)))
print("set-up (frame, renderers) %.2f s, initialisation %.2f s" % (t1 - t0, t2 - t1))
print("fit: %d iterations in %.2f s (%.1f ms per iteration), logL %.1f -> %.1f"
      % (n, t3 - t2, 1e3 * (t3 - t2) / n, -blend.loss[0], logL))
blend2 = scarlet.Blend(sources, observations)
t4 = time.time()
n2, logL2 = blend2.fit(n_iter, e_rel=1e-4)
t5 = time.time()
print("second fit (operators resident): %d iterations in %.2f s (%.1f ms per iteration), logL %.1f"
      % (n2, t5 - t4, 1e3 * (t5 - t4) / max(n2, 1), logL2))
