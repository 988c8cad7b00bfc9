"""Parameterisation of ``scarlet.lite`` components and the joint spectrum fit
(reference scarlet/lite/initialization.py:140-186, 250-318, 608-645).

``init_all_sources_main`` (lite/initialization.py:321-419) is provided with both
monotonicity variants (weighted sweep, or ``use_mask=True``: the monotonic mask operators);
the wavelet initialisation (``init_all_sources_wavelets``) needs the starlet transform and
is not part of this package.
"""

from functools import partial

import numpy as np

from ..bbox import Box, overlapped_slices
from ..initialization import trim_morphology
from ..operator import prox_monotonic_mask, prox_uncentered_symmetry, prox_weighted_monotonic
from ..parameter import relative_step
from .measure import calculate_snr
from .models import LiteComponent, LiteFactorizedComponent, LiteSource
from .parameters import AdaproxParameter, FistaParameter
from .utils import bounds_to_bbox, insert_image, project_morph_to_center


def get_min_psf(psfs, thresh=0.01):
    """Central part of a (bands, h, w) PSF cube outside of which no two bands differ by
    more than ``thresh`` relative to their overall maximum (lite/initialization.py:19-80)."""
    py, px = psfs.shape[1] // 2, psfs.shape[2] // 2
    xx, yy = np.meshgrid(np.arange(psfs.shape[-1]), np.arange(psfs.shape[-2]))
    R = np.sqrt((xx - px) ** 2 + (yy - py) ** 2)
    max_radius = 0
    for p1 in range(len(psfs) - 1):
        for p2 in range(p1 + 1, len(psfs)):
            diff = (psfs[p1] - psfs[p2]) / np.max([psfs[p1], psfs[p2]])
            max_radius = max(max_radius, int(np.max(R * (np.abs(diff) > thresh))))
    dy, dx = py - max_radius, px - max_radius
    sy = slice(dy, -dy) if dy > 0 else slice(None)
    sx = slice(dx, -dx) if dx > 0 else slice(None)
    return psfs[:, sy, sx].copy()


def init_monotonic_morph(detect, center, full_box, grow=0, normalize=True, use_mask=True,
                         thresh=0):
    """Morphology of a monotonic source cut out of the 2-D detection image ``detect``:
    the radial monotonicity operator ('angle' weights, sweep on the GPU) centred on
    ``center``, trimmed at ``thresh`` (lite/initialization.py:83-138).  Returns
    ``(bbox, morph)``; ``morph`` is None when nothing is left."""
    if use_mask:
        _, morph, bounds = prox_monotonic_mask(detect, 0, center, max_iter=0)
        bbox = bounds_to_bbox(bounds)
        if bbox.shape == (1, 1) and morph[bbox.slices][0, 0] == 0:
            return bbox, None
        if grow is not None and grow > 0:
            bbox = bbox.grow(grow)
        morph, bbox = project_morph_to_center(morph, center, bbox, full_box)
    else:
        prox = prox_weighted_monotonic(detect.shape, neighbor_weight="angle", center=center,
                                       min_gradient=0)
        morph = prox(detect, 0).reshape(detect.shape)
        morph, bbox = trim_morphology(center, morph, bg_thresh=thresh)
        if np.max(morph) == 0:
            return Box((0, 0, 0)), None
    if normalize:
        morph /= np.max(morph)
    return bbox, morph


def init_main_parameters(detect, center, observation, convolved=None, use_mask=False, thresh=0.5):
    """Box, morphology and spectrum of one source the way scarlet main initialises an
    ExtendedSource (lite/initialization.py:188-247): symmetrised detection image ->
    monotonic morphology trimmed at ``thresh * mean(noise_rms)``; spectrum = data over
    convolved morphology at the centre pixel."""
    symmetric = prox_uncentered_symmetry(detect.copy(), 0, center, "sdss")
    bbox, morph = init_monotonic_morph(symmetric, center, observation.bbox[1:], grow=0,
                                       normalize=False, use_mask=use_mask,
                                       thresh=np.mean(observation.noise_rms) * thresh)
    if morph is None:
        return bbox, None, None
    images = observation.images
    at_center = (slice(None), center[0], center[1])
    if convolved is None:
        full = insert_image(observation.bbox[1:], bbox, morph)
        convolved = observation.convolve(np.repeat(full[None], images.shape[0], axis=0), mode="real")
    sed = images[at_center] / convolved[at_center]
    sed[sed < 0] = 0
    peak = np.max(morph)
    return bbox, morph / peak, sed * peak


def init_all_sources_main(observation, centers, detect=None, min_snr=50, use_mask=False,
                          percentile=25, thresh=0.5):
    """One ``LiteSource`` of plain ``LiteComponent``s per centre
    (lite/initialization.py:321-419): PSF-shaped if nothing monotonic is found, two
    components (bulge above / disk below ``percentile`` % of the peak, spectra by a joint
    fit) when the PSF-weighted SNR allows ``2 * min_snr``, otherwise one.  Wrap the
    result with ``parameterize_sources``."""
    if detect is None:
        detect = np.sum(observation.images / (observation.noise_rms**2)[:, None, None], axis=0)
    bands = observation.shape[0]
    convolved = observation.convolve(np.repeat(detect[None], bands, axis=0), mode="real")
    model_psf = observation.model_psf[0]
    py, px = model_psf.shape[0] // 2, model_psf.shape[1] // 2
    psf_sed = observation.convolve(np.repeat(observation.model_psf, bands, axis=0),
                                   mode="real")[:, py, px]
    spec_box = observation.bbox[0]
    sources = []
    for center in centers:
        snr = np.floor(calculate_snr(observation.images, observation.variance, observation.psfs,
                                     center))
        bbox, morph, sed = init_main_parameters(detect, center, observation, convolved, use_mask,
                                                thresh)
        if morph is None:
            sed = observation.images[:, center[0], center[1]] / psf_sed
            sed[sed < 0] = 0
            bbox = Box(model_psf.shape, origin=(center[0] - py, center[1] - px))
            comps = [LiteComponent(center, spec_box @ bbox, sed, model_psf / np.max(model_psf))]
        elif snr / min_snr >= 2:
            level = percentile / 100
            bulge = np.maximum(morph - level, 0)
            disk = np.minimum(morph, level)
            bulge /= np.max(bulge)
            disk /= np.max(disk)
            bulge_sed, disk_sed = multifit_seds(observation, [bulge, disk], [bbox, bbox])
            comps = [LiteComponent(center, spec_box @ bbox, bulge_sed, bulge),
                     LiteComponent(center, spec_box @ bbox, disk_sed, disk)]
        else:
            comps = [LiteComponent(center, spec_box @ bbox, sed, morph)]
        sources.append(LiteSource(comps, observation.dtype))
    return sources


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
