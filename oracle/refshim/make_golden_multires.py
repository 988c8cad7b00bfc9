"""Golden vectors for BASELINE config 5 (multi-resolution rendering).

BUILD-CONTAINER TOOLING, run with the conda interpreter that has astropy:

    /opt/conda/bin/python3.9 oracle/refshim/make_golden_multires.py

It imports the read-only reference (autograd / proxmin shims only; the real astropy is
needed to unpickle the WCS objects of the test data), replays the set-up of
tests/test_multiresolution.py for every pair of the five images in
data/test_resampling/Multiresolution_tests.npz and both coverages, and stores the
low-resolution renderings of the high-resolution image together with the inputs (images,
PSFs and the linear part of every WCS) in tests/golden/multiresolution.npz.
"""
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))

# NumPy aliases the reference still uses
for name, target in (("asscalar", "asarray"), ("alen", "asarray"), ("msort", "sort"),
                     ("sometrue", "any"), ("alltrue", "all"), ("product", "prod"),
                     ("cumproduct", "cumprod"), ("round_", "round"), ("asfarray", "asarray")):
    if not hasattr(np, name):
        setattr(np, name, getattr(np, target))
for name, t in (("float", float), ("int", int), ("bool", bool), ("object", object),
                ("complex", complex), ("str", str)):
    if name not in np.__dict__:
        setattr(np, name, t)

# only the autograd / proxmin shims: astropy must be the real one
shim_dir = tempfile.mkdtemp()
for pkg in ("autograd", "proxmin"):
    os.symlink(os.path.join(HERE, "shims", pkg), os.path.join(shim_dir, pkg))
sys.path[:0] = [shim_dir, REPO, "/root/reference"]
for modname in ("scarlet.operators_pybind11", "scarlet.detect_pybind11"):
    sys.modules[modname] = types.ModuleType(modname)
for f in ("prox_weighted_monotonic", "apply_filter", "get_valid_monotonic_pixels",
          "linear_interpolate_invalid_pixels"):
    setattr(sys.modules["scarlet.operators_pybind11"], f, None)
for f in ("get_footprints", "get_connected_pixels", "get_connected_multipeak"):
    setattr(sys.modules["scarlet.detect_pybind11"], f, None)
import scarlet  # noqa: E402

d = np.load("/root/reference/data/test_resampling/Multiresolution_tests.npz", allow_pickle=True)
images, psfs, wcss = d["images"], d["psf"], d["wcs"]
out = dict(n=len(images))
for k, (im, psf, w) in enumerate(zip(images, psfs, wcss)):
    w.array_shape = w.wcs.crpix * 2  # as tests/test_multiresolution.py:64
    out["image_%d" % k] = np.asarray(im)
    out["psf_%d" % k] = np.asarray(psf)
    out["crpix_%d" % k] = np.array(w.wcs.crpix)
    out["crval_%d" % k] = np.array(w.wcs.crval)
    out["pc_%d" % k] = np.array(w.wcs.pc)
    out["cdelt_%d" % k] = np.array(w.wcs.cdelt)


def setup(i, j, coverage):
    obs_hr = scarlet.Observation(images[i][None], wcs=wcss[i], psf=scarlet.ImagePSF(psfs[i]),
                                 channels=["lr"])
    obs_lr = scarlet.Observation(images[j][None], wcs=wcss[j], psf=scarlet.ImagePSF(psfs[j]),
                                 channels=["hr"])
    frame = scarlet.Frame.from_observations([obs_lr, obs_hr], obs_id=1, coverage=coverage)
    return obs_lr, obs_hr, frame


pairs = []
for i in range(len(images)):
    for j in range(i + 1, len(images)):
        for coverage in ("union", "intersection"):
            obs_lr, obs_hr, frame = setup(i, j, coverage)
            r = obs_lr.renderer
            assert type(r).__name__ == "ResolutionRenderer" and not r.isrot
            rendered = obs_lr.render(images[i][None])
            tag = "%d_%d_%s" % (i, j, coverage)
            out["frame_shape_" + tag] = np.array(frame.shape)
            out["frame_crpix_" + tag] = np.array(frame.wcs.wcs.crpix)
            out["fft_shape_" + tag] = np.array(r._fft_shape)
            out["rendered_" + tag] = rendered
            out["hr_renderer_" + tag] = type(obs_hr.renderer).__name__
            out["model_psf_" + tag] = np.asarray(frame.psf.get_model())
            pairs.append(tag)
            sdr = 10 * np.log10(np.sum(rendered ** 2) ** 0.5 / np.sum((rendered - images[j]) ** 2) ** 0.5)
            print(tag, frame.shape, r._fft_shape, "SDR %.1f" % sdr)
out["pairs"] = np.array(pairs)
np.savez_compressed(os.path.join(REPO, "tests", "golden", "multiresolution.npz"), **out)
print("wrote multiresolution.npz")
