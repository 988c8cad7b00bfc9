"""Does the reference's ResolutionRenderer accept a low-resolution observation that is not
square (the ``small_axis = False`` branch, renderer.py:354-363, 536-545)?

BUILD-CONTAINER TOOLING, run with the conda interpreter that has astropy:

    /opt/conda/bin/python3.9 oracle/refshim/check_nonsquare_lowres.py

It does not: renderer.py:274 stacks the two pixel ranges ``np.arange(Ny)`` and
``np.arange(Nx)`` as columns of one array, which raises ``ValueError: all input arrays must
have the same shape`` for every Ny != Nx -- taller or wider -- and a square observation has
``small_axis = Nx <= Ny = True``.  The other unrotated branch is therefore unreachable in
the reference, and scarlet_amd.ResolutionRenderer has no counterpart for it.
Output of this script on the test data (images 3 -> crops of image 4):
    (28, 38) ValueError all input arrays must have the same shape
    (38, 28) ValueError all input arrays must have the same shape
"""
import copy
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conda_reference  # noqa: E402

scarlet = conda_reference.load()
d = np.load("/root/reference/data/test_resampling/Multiresolution_tests.npz", allow_pickle=True)
images, psfs, wcss = d["images"], d["psf"], d["wcs"]
for w in wcss:
    w.array_shape = w.wcs.crpix * 2
i, j = 3, 4
for crop in ((slice(10, 38), slice(None)), (slice(None), slice(10, 38))):
    im = images[j][crop]
    w = copy.deepcopy(wcss[j])
    w.wcs.crpix = w.wcs.crpix - np.array([crop[1].start or 0, crop[0].start or 0])
    w.array_shape = im.shape
    try:
        obs_hr = scarlet.Observation(images[i][None], wcs=wcss[i], psf=scarlet.ImagePSF(psfs[i]),
                                     channels=["lr"])
        obs_lr = scarlet.Observation(im[None], wcs=w, psf=scarlet.ImagePSF(psfs[j]), channels=["hr"])
        scarlet.Frame.from_observations([obs_lr, obs_hr], obs_id=1, coverage="union")
        print(im.shape, "small_axis =", obs_lr.renderer.small_axis,
              obs_lr.render(images[i][None]).shape)
    except ValueError as e:
        print(im.shape, type(e).__name__, e)
