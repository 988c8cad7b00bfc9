"""Parameterisation of ``scarlet.lite`` components and the joint spectrum fit
(reference scarlet/lite/initialization.py:140-186, 250-318, 608-645).

The detection-image initialisation (``init_all_sources_main`` / ``_wavelets``) relies on
the monotonic mask operators and the starlet transform and is not part of this package;
start from spectra / morphologies obtained elsewhere (e.g. ``scarlet_amd.initialization``)
and wrap them with ``init_adaprox_component`` / ``init_fista_component``.
"""

from functools import partial

import numpy as np

from ..bbox import overlapped_slices
from ..parameter import relative_step
from .models import LiteFactorizedComponent, LiteSource
from .parameters import AdaproxParameter, FistaParameter
from .utils import insert_image


def multifit_seds(observation, morphs, boxes):
    """Least-squares spectra of all components at once, band by band: the convolved
    morphologies are the columns of the design matrix."""
    if len(morphs) != len(boxes):
        raise ValueError("morphs and boxes should have the same number of parameters, "
                         "got {} and {} respectively".format(len(morphs), len(boxes)))
    bands = observation.images.shape[0]
    dtype = observation.images.dtype
    spec_box = observation.bbox[0]
    full_box = boxes[0]
    for box in boxes[1:]:
        full_box = full_box | box
    full_box = spec_box @ full_box
    img = insert_image(full_box, observation.bbox, observation.images)
    design = np.zeros((bands, len(morphs), img[0].size), dtype=dtype)
    for k, (morph, bbox) in enumerate(zip(morphs, boxes)):
        # one broadcast copy of the morphology per band, convolved with that band's kernel
        cube = np.repeat(insert_image(full_box[1:], bbox, morph)[None], bands, axis=0)
        design[:, k] = observation.convolve(cube).reshape(bands, -1)
    seds = np.zeros((len(morphs), bands), dtype=dtype)
    for b in range(bands):
        seds[:, b] = np.linalg.lstsq(design[b].T, img[b].reshape(-1), rcond=None)[0]
    seds[seds < 0] = 0
    return seds


def init_adaprox_component(center, bbox, sed, morph, observation, factor=10, bg_thresh=None,
                           max_prox_iter=1):
    """Component whose parameters follow proximal AMSGrad: spectrum step 1 % of its mean
    but at least ``noise_rms / factor``, morphology step 1e-2."""
    sed = AdaproxParameter(
        sed, step=partial(relative_step, factor=1e-2, minimum=observation.noise_rms / factor),
        max_prox_iter=max_prox_iter)
    morph = AdaproxParameter(morph, step=1e-2, max_prox_iter=max_prox_iter)
    return LiteFactorizedComponent(sed, morph, center, bbox, observation.bbox,
                                   observation.noise_rms, bg_thresh=bg_thresh)


def init_fista_component(center, bbox, sed, morph, observation, bg_thresh=None):
    """Component whose parameters follow FISTA with step 1 / (2 <w>), <w> the mean
    positive weight inside the box."""
    _, in_obs = overlapped_slices(bbox, observation.bbox)
    w = observation.weights[in_obs]
    step = 1 / (2 * np.mean(w[w > 0]))
    return LiteFactorizedComponent(FistaParameter(sed, step=step), FistaParameter(morph, step=step),
                                   center, bbox, observation.bbox, observation.noise_rms,
                                   bg_thresh=bg_thresh)


def parameterize_sources(sources, observation, parameterization):
    """Re-wrap the spectra / morphologies of ``sources`` with ``parameterization``
    (e.g. ``init_adaprox_component``); inputs are copied."""
    out = []
    for src in sources:
        comps = [parameterization(center=tuple(c.center), sed=c.sed.copy(), morph=c.morph.copy(),
                                  bbox=c.bbox.copy(), observation=observation)
                 for c in src.components]
        out.append(LiteSource(comps, src.dtype))
    return out
