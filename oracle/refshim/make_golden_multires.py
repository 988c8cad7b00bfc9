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

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conda_reference  # noqa: E402

scarlet = conda_reference.load()
REPO = conda_reference.REPO

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


# ---------------------------------------------------------------------------------------
# A two-observation blend (high-resolution image 3 + low-resolution image 4, union frame)
# evaluated by the reference in float64: model, both renderings, both log-likelihoods and
# central finite differences of the total -logL with respect to every parameter.  Pins
# the multi-resolution term of the fit (loss and gradient) in tests/golden/multires_fit.npz.
# ---------------------------------------------------------------------------------------
def fit_scene():
    i_hr, i_lr = 3, 4
    rng = np.random.default_rng(77)
    im_hr = np.asarray(images[i_hr], dtype=np.float64)[None]
    im_lr = np.asarray(images[i_lr], dtype=np.float64)[None]
    w_hr = np.full(im_hr.shape, 1 / (0.02 * im_hr.max()) ** 2)
    w_lr = np.full(im_lr.shape, 1 / (0.02 * im_lr.max()) ** 2)
    w_hr[0, 5:8, 30:34] = 0  # masked pixels
    w_lr[0, 20, 3:9] = 0
    obs_hr = scarlet.Observation(im_hr, wcs=wcss[i_hr], psf=scarlet.ImagePSF(psfs[i_hr]),
                                 channels=["hr"], weights=w_hr)
    obs_lr = scarlet.Observation(im_lr, wcs=wcss[i_lr], psf=scarlet.ImagePSF(psfs[i_lr]),
                                 channels=["lr"], weights=w_lr)
    observations = [obs_lr, obs_hr]
    frame = scarlet.Frame.from_observations(observations, obs_id=1, coverage="union")
    frame.dtype = np.float64  # evaluate in double so that finite differences are clean
    for obs in observations:
        obs.match(frame)
    r_lr, r_hr = obs_lr.renderer, obs_hr.renderer
    assert type(r_lr).__name__ == "ResolutionRenderer" and type(r_hr).__name__ == "ConvolutionRenderer"
    assert r_lr.small_axis and not r_lr.isrot

    C, H, W = frame.shape
    yy, xx = np.mgrid[:15, :15] - 7.0
    sources, spec = [], []
    for (cy, cx), sigma, q in (((26, 26), 2.5, 0.8), ((17, 31), 2.0, 1.0), ((34, 19), 3.0, 0.6)):
        image = np.exp(-(yy**2 / q + xx**2 * q) / (2 * sigma**2))
        image *= rng.uniform(0.9, 1.1, image.shape)
        image /= image.max()
        sed = np.array([0.7 * im_lr.max(), 0.2 * im_hr.max()]) * rng.uniform(0.5, 1.5, 2)
        spectrum = scarlet.TabulatedSpectrum(frame, sed.copy())
        morphology = scarlet.ImageMorphology(
            frame, image.copy(), bbox=scarlet.Box((15, 15), origin=(cy - 7, cx - 7)))
        sources.append(scarlet.FactorizedComponent(frame, spectrum, morphology))
        spec.append((sed, image, (cy - 7, cx - 7)))
    blend = scarlet.Blend(sources, observations)

    def neg_logL(parameters):
        model = blend.get_model(*parameters)
        return -sum(obs.get_log_likelihood(model) for obs in observations)

    X = [np.array(p, dtype=np.float64) for p in blend.parameters]
    model = blend.get_model(*X)
    out = dict(
        channels=np.array(frame.channels), frame_shape=np.array(frame.shape),
        frame_crpix=np.array(frame.wcs.wcs.crpix),
        model_psf=np.asarray(frame.psf.get_model()),
        data_hr=obs_hr.data, weights_hr=obs_hr.weights, data_lr=obs_lr.data,
        weights_lr=obs_lr.weights, i_hr=i_hr, i_lr=i_lr,
        model=model, rendered_hr=obs_hr.render(model), rendered_lr=obs_lr.render(model),
        logL_hr=obs_hr.get_log_likelihood(model), logL_lr=obs_lr.get_log_likelihood(model),
        log_norm_hr=obs_hr.log_norm, log_norm_lr=obs_lr.log_norm,
        # set-up quantities of the two renderers
        hr_kernel=r_hr.diff_kernel.image, hr_slices=np.array(
            [[s.start, s.stop] for sl in r_hr.slices for s in sl[-2:]]),
        lr_kernel=r_lr.diff_kernel.image, lr_shifts=r_lr.shifts, lr_other_shifts=r_lr.other_shifts,
        lr_h=r_lr.h, lr_fft_shape=np.array(r_lr._fft_shape),
        n_components=len(spec),
    )
    for k, (sed, image, origin) in enumerate(spec):
        out["sed_%d" % k], out["morph_%d" % k], out["origin_%d" % k] = sed, image, np.array(origin)
    # central differences, parameter order of blend.parameters (spectrum, image, shift)
    names = [p.name for p in blend.parameters]
    out["parameter_names"] = np.array(names)
    for j, (p, name) in enumerate(zip(X, names)):
        if name == "shift":
            continue
        g = np.zeros(p.shape)
        flat, gflat = p.reshape(-1), g.reshape(-1)
        for e in range(flat.size):
            keep = flat[e]
            eps = 1e-5 * max(abs(keep), 1e-2)
            flat[e] = keep + eps
            up = neg_logL(X)
            flat[e] = keep - eps
            down = neg_logL(X)
            flat[e] = keep
            gflat[e] = (up - down) / (2 * eps)
        out["fd_%d" % j] = g
        print("fd", j, name, float(np.abs(g).max()))
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "multires_fit.npz"), **out)
    print("wrote multires_fit.npz: logL_hr %.6f logL_lr %.6f" % (out["logL_hr"], out["logL_lr"]))


fit_scene()
