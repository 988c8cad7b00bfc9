"""Extended-source models (reference scarlet/source.py:249-522, 615-807):
``ExtendedSource`` factory -> ``SingleExtendedSource`` / ``MultiExtendedSource`` /
``CompactExtendedSource``, and ``PointSource`` (source.py:92-128).  Gaussian,
Spergel, starlet and random sources are outside the scope of this package."""

import logging

import numpy as np

from . import initialization as init
from . import operator
from .bbox import Box, overlapped_slices
from .component import CombinedComponent, FactorizedComponent
from .constraint import CenterOnConstraint
from .morphology import ExtendedSourceMorphology, PointSourceMorphology
from .parameter import Parameter
from .spectrum import TabulatedSpectrum

logger = logging.getLogger("scarlet_amd.source")


def _noise_rms(observations):
    return np.concatenate(
        [np.array(np.mean(obs.noise_rms, axis=(1, 2))) for obs in observations]
    ).reshape(-1)


class PointSource(FactorizedComponent):
    """Point source: the model PSF at a free centre times a spectrum taken from the
    peak pixel of the observations, corrected for the PSF (source.py:92-128)."""

    def __init__(self, model_frame, sky_coord, observations):
        if not hasattr(observations, "__iter__"):
            observations = (observations,)
        center = Parameter(np.array(model_frame.get_pixel(sky_coord), dtype=float),
                           name="center", step=3e-2)
        morphology = PointSourceMorphology(model_frame, center)
        spectrum = init.get_pixel_spectrum(sky_coord, observations, correct_psf=True)
        spectrum = TabulatedSpectrum(model_frame, spectrum, min_step=_noise_rms(observations))
        super().__init__(model_frame, spectrum, morphology)
        self.center = morphology.center


class CompactExtendedSource(FactorizedComponent):
    """Source initialised with the model PSF as morphology."""

    def __init__(self, model_frame, sky_coord, observations, shifting=False, resizing=True,
                 boxsize=None):
        if not hasattr(observations, "__iter__"):
            observations = (observations,)
        assert model_frame.psf is not None
        morph, bbox = self.init_morph(model_frame, sky_coord, boxsize=boxsize)
        center = model_frame.get_pixel(sky_coord)
        morphology = ExtendedSourceMorphology(model_frame, center, morph, bbox=bbox,
                                              monotonic="angle", symmetric=False, min_grad=0,
                                              shifting=shifting, resizing=resizing)
        spectrum = init.get_pixel_spectrum(sky_coord, observations, correct_psf=True)
        spectrum /= morph.sum()
        spectrum = TabulatedSpectrum(model_frame, spectrum, min_step=_noise_rms(observations))
        super().__init__(model_frame, spectrum, morphology)
        self.center = morphology.center

    @staticmethod
    def init_morph(frame, sky_coord, boxsize=None):
        """Band-averaged model PSF, peak-normalised, in a standard-size box."""
        ci = np.round(frame.get_pixel(sky_coord)).astype("int")
        psf = frame.psf.get_model().mean(axis=0)
        psf_box = Box(psf.shape, origin=(ci[0] - psf.shape[0] // 2, ci[1] - psf.shape[1] // 2))
        if boxsize is None:
            boxsize = init.get_minimal_boxsize(max(psf.shape))
        morph = np.zeros((boxsize, boxsize))
        bbox = Box(morph.shape, origin=(ci[0] - boxsize // 2, ci[1] - boxsize // 2))
        dst, src = overlapped_slices(bbox, psf_box)
        morph[dst] = psf[src]
        morph /= morph.max()
        return morph, bbox


class SingleExtendedSource(FactorizedComponent):
    """One spectrum x one monotonic morphology, initialised from the
    signal-to-noise weighted coadd of the observations."""

    def __init__(self, model_frame, sky_coord, observations, thresh=1.0, shifting=False,
                 resizing=True, boxsize=None):
        if not hasattr(observations, "__iter__"):
            observations = (observations,)
        spectra = init.get_pixel_spectrum(sky_coord, observations, concat=False)
        spectrum = TabulatedSpectrum(model_frame, np.concatenate(spectra).reshape(-1),
                                     min_step=_noise_rms(observations))
        image, std = init.build_initialization_image(observations, spectra=spectra)
        morph, bbox = self.init_morph(model_frame, sky_coord, image, std, thresh=thresh,
                                      symmetric=True, monotonic="flat", min_grad=0,
                                      boxsize=boxsize)
        center = model_frame.get_pixel(sky_coord)
        morphology = ExtendedSourceMorphology(model_frame, center, morph, bbox=bbox,
                                              monotonic="angle", symmetric=False, min_grad=0,
                                              shifting=shifting, resizing=resizing)
        super().__init__(model_frame, spectrum, morphology)
        self.center = morphology.center

    @staticmethod
    def init_morph(frame, sky_coord, detect, detect_std, thresh=1, symmetric=True,
                   monotonic="flat", min_grad=0, boxsize=None):
        """Symmetric, monotonic cut-out of the detection image around the source."""
        ci = np.round(frame.get_pixel(sky_coord)).astype("int")
        im = detect.copy()
        if symmetric:
            im = operator.prox_uncentered_symmetry(im, 0, center=ci, algorithm="sdss")
        if monotonic:
            if monotonic is True:
                monotonic = "angle"
            prox = operator.prox_weighted_monotonic(im.shape, neighbor_weight=monotonic,
                                                    center=ci, min_gradient=min_grad)
            im = prox(np.ascontiguousarray(im), 0).reshape(im.shape)
        morph, bbox = init.trim_morphology(ci, im, bg_thresh=detect_std * thresh, boxsize=boxsize)
        if morph.sum() > 0:
            morph /= morph.max()
        else:
            logger.warning(f"No flux in morphology model for source at {sky_coord}")
            morph = CenterOnConstraint(tiny=1)(morph, 0)
        if frame.psf is not None:
            # noisy initialisations leave a few pixels only: never narrower than the PSF
            psf_morph, _ = CompactExtendedSource.init_morph(frame, sky_coord,
                                                            boxsize=max(bbox.shape))
            morph = np.maximum(morph, psf_morph)
        return morph, bbox


class MultiExtendedSource(CombinedComponent):
    """K components stacked vertically: the single-source morphology is cut at
    flux percentiles into an outer plateau and inner remainders."""

    def __init__(self, model_frame, sky_coord, observations, K=2, flux_percentiles=None,
                 thresh=1.0, shifting=False, resizing=True, boxsize=None):
        if flux_percentiles is None:
            flux_percentiles = (25,)
        assert K == len(flux_percentiles) + 1
        if not hasattr(observations, "__iter__"):
            observations = (observations,)
        base = ExtendedSource(model_frame, sky_coord, observations, thresh=thresh, boxsize=boxsize)
        spectrum, morphology = base.children
        spectrum = spectrum.get_parameter(0)._data
        morphs, boxes = self.init_morphs(morphology, flux_percentiles)
        center = model_frame.get_pixel(sky_coord)
        noise_rms = _noise_rms(observations)
        components = []
        for k in range(K):
            spec = TabulatedSpectrum(model_frame, spectrum.copy(), min_step=noise_rms / 10)
            morph = ExtendedSourceMorphology(model_frame, center, morphs[k], bbox=boxes[k],
                                             monotonic="angle", symmetric=False, min_grad=0,
                                             shifting=shifting, resizing=resizing)
            self.center = morph.center
            components.append(FactorizedComponent(model_frame, spec, morph))
        super().__init__(components)

    @staticmethod
    def init_morphs(morphology, flux_percentiles):
        morph = morphology.get_model()
        K = len(flux_percentiles) + 1
        layers = np.zeros((K,) + morph.shape, dtype=morph.dtype)
        layers[0] = morph
        peak = morph.max()
        below = 0
        for k, perc in enumerate(np.sort(flux_percentiles), start=1):
            cut = perc * peak / 100
            inside = morph > cut
            layers[k - 1][inside] = cut - below
            layers[k][inside] = morph[inside] - cut
            below = cut
        for k in range(K):
            if np.all(layers[k] <= 0):
                logger.warning(f"Zero or negative morphology for component {k}")
            layers[k] /= layers[k].max()
        return layers, tuple(morphology.bbox.copy() for _ in range(K))


def ExtendedSource(model_frame, sky_coord, observations, K=1, flux_percentiles=None, thresh=1.0,
                   compact=False, shifting=False, resizing=True, boxsize=None):
    """Factory: ``CompactExtendedSource`` if ``compact``, ``SingleExtendedSource`` for
    ``K == 1``, else ``MultiExtendedSource`` with ``K`` components."""
    if compact:
        return CompactExtendedSource(model_frame, sky_coord, observations, shifting=shifting,
                                     resizing=resizing, boxsize=boxsize)
    if K == 1:
        return SingleExtendedSource(model_frame, sky_coord, observations, thresh=thresh,
                                    shifting=shifting, resizing=resizing, boxsize=boxsize)
    return MultiExtendedSource(model_frame, sky_coord, observations, K=K,
                               flux_percentiles=flux_percentiles, thresh=thresh,
                               shifting=shifting, resizing=resizing, boxsize=boxsize)
