"""Seeded synthetic blends for the benchmark configurations (SURVEY.md 8d,
BASELINE.md config 2/3): C=5 bands, 128x128 pixels, K=10 extended sources with
elliptical-Gaussian morphologies in 41x41 boxes, band-shared Gaussian observed
PSF (sigma 2.0, 41x41 stamp), model PSF GaussianPSF(0.8), white noise 0.05
(weights 400).  ``make_blend(seed)`` is deterministic in NumPy's PCG64 stream.
"""

import numpy as np

from . import fft
from .psf import GaussianPSF

C, H, W, K, BOX = 5, 128, 128, 10, 41
SIGMA_OBS, SIGMA_MODEL, NOISE = 2.0, 0.8, 0.05


def psfs():
    """(observed PSF (1,41,41), model PSF (1,9,9), difference kernel (1,41,41)), float32."""
    obs = GaussianPSF(SIGMA_OBS, boxsize=BOX).get_model().astype(np.float32)
    model = GaussianPSF(SIGMA_MODEL).get_model().astype(np.float32)
    diff = fft.match_psf(fft.Fourier(obs), fft.Fourier(model), padding=10)
    return obs, model, diff.image.astype(np.float32)


def _place(cube, sed, morph, origin):
    y0, x0 = origin
    h, w = morph.shape
    ylo, yhi = max(y0, 0), min(y0 + h, cube.shape[1])
    xlo, xhi = max(x0, 0), min(x0 + w, cube.shape[2])
    cube[:, ylo:yhi, xlo:xhi] += (
        sed[:, None, None] * morph[None, ylo - y0 : yhi - y0, xlo - x0 : xhi - x0]
    )


def _draw_truth(rng, n_sources):
    centers = rng.integers(20, 108, size=(n_sources, 2))
    yy, xx = np.mgrid[:BOX, :BOX] - BOX // 2
    true_seds, true_morphs, origins = [], [], []
    for k in range(n_sources):
        sigma, q, theta = rng.uniform(2, 5), rng.uniform(0.5, 1), rng.uniform(0, np.pi)
        u = np.cos(theta) * xx + np.sin(theta) * yy
        v = -np.sin(theta) * xx + np.cos(theta) * yy
        morph = np.exp(-0.5 * (u**2 + (v / q) ** 2) / sigma**2)
        true_morphs.append((morph / morph.max()).astype(np.float32))
        true_seds.append(rng.uniform(1, 5, size=C).astype(np.float32))
        origins.append(centers[k] - BOX // 2)
    return true_seds, true_morphs, origins


def _finish(rng, rendered, true_seds, true_morphs, origins, kernel):
    obs_psf, model_psf, diff = kernel
    data = (rendered + rng.normal(0, NOISE, size=rendered.shape)).astype(np.float32)
    weights = np.full(rendered.shape, 1 / NOISE**2, dtype=np.float32)
    seds, morphs = [], []
    for sed, morph in zip(true_seds, true_morphs):
        seds.append((sed * rng.uniform(0.7, 1.3, size=C)).astype(np.float32))
        m = morph * rng.uniform(0.8, 1.2, size=morph.shape)
        morphs.append((m / m.max()).astype(np.float32))
    return dict(
        data=data, weights=weights, true_seds=np.array(true_seds), true_morphs=true_morphs,
        seds=np.array(seds), morphs=morphs, origins=np.array(origins, dtype=np.int64),
        obs_psf=obs_psf, model_psf=model_psf, diff_kernel=diff,
        noise_rms=np.full(C, NOISE, dtype=np.float32),
    )


def make_blend(seed=1234, kernel=None, n_sources=K):
    """One synthetic scene, rendered on the host.  Returns a dict with ``data``,
    ``weights`` (C,H,W) float32, the truth (``true_seds``, ``true_morphs``), the
    perturbed initial parameters (``seds`` (K,C) float32, ``morphs`` list of
    (41,41) float32, ``origins`` (K,2) int), ``obs_psf``, ``model_psf``,
    ``diff_kernel`` and ``noise_rms`` (C,)."""
    rng = np.random.default_rng(seed)
    kernel = psfs() if kernel is None else kernel
    true_seds, true_morphs, origins = _draw_truth(rng, n_sources)
    truth = np.zeros((C, H, W), dtype=np.float32)
    for sed, morph, origin in zip(true_seds, true_morphs, origins):
        _place(truth, sed, morph, origin)
    rendered = fft.convolve(fft.Fourier(truth), fft.Fourier(kernel[2]), axes=(1, 2)).image
    return _finish(rng, rendered, true_seds, true_morphs, origins, kernel)


def make_batch(seeds, kernel=None, n_sources=K, device=0, chunk=256):
    """The same scenes as ``make_blend`` for many seeds, with the noiseless truth
    rendered on the GPU (model render + PSF convolution of the device path), so
    that setting up a 1024-blend benchmark takes seconds.  Same random stream per
    seed as ``make_blend``; data agree to float32 rounding."""
    from .batch import BlendBatch, ComponentSpec

    kernel = psfs() if kernel is None else kernel
    seeds = list(seeds)
    rngs = [np.random.default_rng(s) for s in seeds]
    truths = [_draw_truth(r, n_sources) for r in rngs]
    scenes = []
    zeros = np.zeros((min(chunk, len(seeds)), C, H, W), dtype=np.float32)
    for lo in range(0, len(seeds), chunk):
        part = truths[lo : lo + chunk]
        comps = [
            [ComponentSpec(sed, morph, origin, prox_flags=0)
             for sed, morph, origin in zip(*t)]
            for t in part
        ]
        z = zeros[: len(part)]
        batch = BlendBatch(z, z + 1, comps, kernel=kernel[2], max_iter=1, device=device)
        _, rendered, _ = batch.forward(model=False)
        batch.close()
        for i, t in enumerate(part):
            scenes.append(_finish(rngs[lo + i], rendered[i], *t, kernel))
    return scenes
