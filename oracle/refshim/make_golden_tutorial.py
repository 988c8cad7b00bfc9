"""Golden vectors for the scene of the reference's multi-resolution tutorial.

BUILD-CONTAINER TOOLING, run with the conda interpreter that has astropy:

    /opt/conda/bin/python3.9 oracle/refshim/make_golden_tutorial.py

docs/tutorials/multiresolution.ipynb: a 5-band HSC cut-out (50 x 50 pixels of 0.168") and
an HST F814W cut-out (250 x 250 pixels of 0.03") of the same field, their PSFs, a model
frame on the HST grid with a Gaussian model PSF (``Frame.from_observations(...,
coverage="intersection", model_psf=GaussianPSF(0.6))``), ``ExtendedSource``s initialised
from both observations, ``set_spectra_to_match``.  The tutorial finds its sources with
``sep`` on a wavelet detection image (neither is in scope here); this script takes the
brightest well-separated peaks of the HST image instead.  The reference cannot run
``Blend.fit`` in this container (no autograd / proxmin), so the golden holds everything up
to the fit: the input data (MIT-licensed test data of the reference) with the celestial
WCS keywords, astropy's pixel <-> sky conversions at sample points (pins the facade's
gnomonic WCS), the model frame, the renderers' set-up, the initialised sources, both
renderings of the initial model and both log-likelihoods.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conda_reference  # noqa: E402

scarlet = conda_reference.load()
import astropy.io.fits as fits  # noqa: E402
from astropy.wcs import WCS  # noqa: E402

DATA = "/root/reference/data/test_resampling/"


def native(a, dtype=np.float32):
    return np.ascontiguousarray(np.asarray(a).astype(dtype))


# -- cells 2-3 of the notebook -----------------------------------------------------------
obs_hdu = fits.open(DATA + "Cut_HSC1.fits")
data_hsc = native(obs_hdu[0].data)
wcs_hsc = WCS(obs_hdu[0].header)
channels_hsc = ["g", "r", "i", "z", "y"]
psf_hsc_image = native(fits.open(DATA + "PSF_HSC.fits")[0].data)
hst_hdu = fits.open(DATA + "Cut_HST1.fits")
data_hst = native(hst_hdu[0].data)[None]
wcs_hst = WCS(hst_hdu[0].header)
channels_hst = ["F814W"]
psf_hst_image = native(fits.open(DATA + "PSF_HST.fits")[0].data)[None]
data_hst *= data_hsc.max() / data_hst.max()

out = dict(data_hsc=data_hsc, data_hst=data_hst, psf_hsc=psf_hsc_image, psf_hst=psf_hst_image,
           channels_hsc=np.array(channels_hsc), channels_hst=np.array(channels_hst))
for tag, w in (("hsc", wcs_hsc), ("hst", wcs_hst)):
    c = w.celestial
    out["crpix_" + tag] = np.array(c.wcs.crpix)
    out["crval_" + tag] = np.array(c.wcs.crval)
    out["pc_" + tag] = np.array(c.wcs.get_pc())
    out["cdelt_" + tag] = np.array(c.wcs.get_cdelt())
    out["ctype_" + tag] = np.array(list(c.wcs.ctype))
    n = 50 if tag == "hsc" else 250
    pix = np.array([[x, y] for x in (-20.0, 0.0, n / 3, n - 1.0, n + 20.0)
                    for y in (-20.0, 0.0, n / 2, n - 1.0, n + 20.0)])
    sky = np.array(c.pixel_to_world_values(pix))
    out["sample_pix_" + tag], out["sample_sky_" + tag] = pix, sky
    out["sample_back_" + tag] = np.array(c.world_to_pixel_values(sky))

obs_hst = scarlet.Observation(data_hst, wcs=wcs_hst, psf=scarlet.ImagePSF(psf_hst_image),
                              channels=channels_hst, weights=None)
obs_hsc = scarlet.Observation(data_hsc, wcs=wcs_hsc, psf=scarlet.ImagePSF(psf_hsc_image),
                              channels=channels_hsc, weights=None)
observations = [obs_hsc, obs_hst]
model_psf = scarlet.GaussianPSF(sigma=0.6)
model_frame = scarlet.Frame.from_observations(observations, coverage="intersection",
                                              model_psf=model_psf)
r_hsc, r_hst = obs_hsc.renderer, obs_hst.renderer
print("frame", model_frame.shape, type(r_hsc).__name__, type(r_hst).__name__,
      getattr(r_hsc, "_fft_shape", None))
out.update(frame_shape=np.array(model_frame.shape), frame_channels=np.array(model_frame.channels),
           frame_crpix=np.array(model_frame.wcs.celestial.wcs.crpix),
           hsc_renderer=type(r_hsc).__name__, hst_renderer=type(r_hst).__name__,
           hsc_fft_shape=np.array(r_hsc._fft_shape), hsc_h=r_hsc.h,
           hsc_shifts=r_hsc.shifts, hsc_other_shifts=r_hsc.other_shifts,
           hst_kernel=native(r_hst.diff_kernel.image),
           hst_slices=np.array([[s.start, s.stop] for sl in r_hst.slices for s in sl[-2:]]))
k = r_hsc.diff_kernel.image
cy, cx = k.shape[1] // 2, k.shape[2] // 2
out["hsc_kernel_center"] = native(k[:, cy - 40:cy + 41, cx - 40:cx + 41])
out["hsc_kernel_sum"] = k.sum(axis=(1, 2))

# -- sources: brightest separated peaks of the HST image instead of the sep catalogue -----
img = data_hst[0]
peaks = []
order = np.argsort(img, axis=None)[::-1]
for flat in order[:20000]:
    y, x = divmod(int(flat), img.shape[1])
    if not (30 <= y < img.shape[0] - 30 and 30 <= x < img.shape[1] - 30):
        continue
    if all((y - py) ** 2 + (x - px) ** 2 > 30**2 for py, px in peaks):
        peaks.append((y, x))
    if len(peaks) == 4:
        break
pixel_hst = np.array(peaks, dtype=float)
ra_dec = obs_hst.get_sky_coord(pixel_hst)
out.update(pixel_hst=pixel_hst, ra_dec=np.array(ra_dec))
print("peaks", peaks)

sources = [scarlet.ExtendedSource(model_frame, sky, observations, thresh=0.1) for sky in ra_dec]
scarlet.initialization.set_spectra_to_match(sources, observations)
blend = scarlet.Blend(sources, observations)
out["n_sources"] = len(sources)
for j, src in enumerate(sources):
    spectrum, morph = src.parameters[0], src.parameters[1]
    out["spectrum_%d" % j] = np.array(spectrum)
    out["morph_%d" % j] = np.array(morph)
    out["origin_%d" % j] = np.array(src.bbox.origin)
    out["shape_%d" % j] = np.array(src.bbox.shape)
    print("source", j, src.bbox, np.array(spectrum))

model = blend.get_model()
rendered_hsc = obs_hsc.render(model)
rendered_hst = obs_hst.render(model)
out.update(model_sum=model.sum(axis=(1, 2)), model_max=model.max(axis=(1, 2)),
           rendered_hsc=native(rendered_hsc), rendered_hst=native(rendered_hst),
           logL_hsc=obs_hsc.get_log_likelihood(model), logL_hst=obs_hst.get_log_likelihood(model))
print("logL", out["logL_hsc"], out["logL_hst"])
path = os.path.join(conda_reference.REPO, "tests", "golden", "multires_tutorial.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path) // 1024, "KiB")
