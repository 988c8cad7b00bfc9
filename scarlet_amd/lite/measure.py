"""Measurements of ``scarlet.lite`` (reference scarlet/lite/measure.py)."""

import numpy as np

from ..bbox import Box, overlapped_slices
from .utils import insert_image


def calculate_snr(images, variance, psfs, center):
    """PSF-weighted signal-to-noise at ``center``: ``sum(I P) / sqrt(sum(P^2 var))`` over
    the PSF stamp placed on the centre."""
    py, px = psfs.shape[1] // 2, psfs.shape[2] // 2
    bbox = Box(psfs.shape, origin=(0, center[0] - py, center[1] - px))
    noise = bbox.extract_from(variance)
    img = bbox.extract_from(images)
    return np.sum(img * psfs) / np.sqrt(np.sum(psfs * noise * psfs))


def weight_sources(blend, mask_footprint=True):
    """Redistribute the observed flux among the sources in proportion to their convolved
    models (the classical deblending template trick): sets ``src.flux`` and
    ``src.flux_box`` on every source (lite/measure.py:39-91)."""
    obs = blend.observation
    py, px = obs.psfs.shape[-2] // 2, obs.psfs.shape[-1] // 2
    images = obs.images.copy()
    if mask_footprint:
        images = images * (obs.weights > 0)
    total = obs.convolve(blend.get_model(), mode="real")
    total[total < 0] = 0
    for src in blend.sources:
        if len(src.components) == 0:
            src.flux = 0
            src.flux_box = Box((0, 0, 0))
            continue
        bbox = src.bbox.grow((0, py, px))
        model = obs.convolve(insert_image(bbox, src.bbox, src.get_model()), mode="real")
        model[model < 0] = 0
        in_obs, in_box = overlapped_slices(obs.bbox, bbox)
        with np.errstate(divide="ignore", invalid="ignore"):
            ratio = model[in_box] / total[in_obs]
        ratio[total[in_obs] == 0] = 0
        ratio[ratio > 1] = 1  # round-off can lift a hot pixel slightly above 1
        src.flux = ratio * images[in_obs]
        src.flux_box = obs.bbox & bbox
