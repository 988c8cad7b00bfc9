"""Point-spread-function models.

Same classes and constructor signatures as the reference (scarlet/psf.py:10-234):
``GaussianPSF(sigma, integrate=True, boxsize=None)``, ``MoffatPSF(alpha=4.7, beta=1.5,
integrate=False, boxsize=None)``, ``ImagePSF(image)``, all with
``get_model(*parameters, offset=None)`` returning a band cube normalised to unit sum and a
``bbox`` centred on the origin.  Set-up time only: PSF images feed the difference-kernel
construction in ``Observation.match`` and the point-source morphology on the host side;
the device evaluates the pixel-integrated Gaussian itself (``point_source_kernel``).
"""

from abc import abstractmethod

import numpy as np
from scipy import special

from .bbox import Box
from .fft import shift
from .model import Model
from .parameter import Parameter, prepare_param


def normalize(image):
    """Unit sum per band, in place."""
    total = image.sum(axis=(1, 2))
    target = image._data if isinstance(image, Parameter) else image
    target /= total[:, None, None]
    return image


def _centered_box(n_bands, height, width):
    """Box of a band cube whose pixel (height // 2, width // 2) sits at the origin."""
    return Box((n_bands, height, width), origin=(0, -(height // 2), -(width // 2)))


class PSF(Model):
    @abstractmethod
    def get_model(self, *parameter, offset=None):
        """Centred PSF image cube, optionally shifted by ``offset`` pixels."""


class FunctionPSF(PSF):
    """PSF given by an analytic profile on an odd ``boxsize`` x ``boxsize`` grid centred
    on the origin.  The first parameter has one entry per band; if all entries agree
    (``is_same``) the model is a single image broadcast over the bands.  Subclasses
    provide ``_profile(Y, X, *band_parameters)``."""

    default_boxsize = 15

    def __init__(self, *parameters, integrate=True, boxsize=None):
        super().__init__(*parameters)
        self.integrate = integrate
        size = self.default_boxsize if boxsize is None else boxsize
        size += 1 - size % 2  # an even size grows by one so that there is a centre pixel
        per_band = self.get_parameter(0, *parameters)
        self.bbox = _centered_box(len(per_band), size, size)
        self._Y = self._X = np.arange(size) - size // 2
        self.is_same = np.all(per_band == per_band[0])
        self._d = self.bbox.D - 2

    def expand_dims(self, model):
        return np.expand_dims(model, axis=tuple(range(self._d)))

    def _evaluate(self, band_parameters, offset):
        """Normalised cube for per-band tuples of profile parameters."""
        oy, ox = (0, 0) if offset is None else offset
        Y, X = self._Y - oy, self._X - ox
        if self.is_same:
            cube = self.expand_dims(self._profile(Y, X, *band_parameters[0]))
        else:
            cube = np.stack([self._profile(Y, X, *p) for p in band_parameters], axis=0)
        return normalize(cube)


class GaussianPSF(FunctionPSF):
    """Circular Gaussian of width ``sigma`` (per band), by default integrated over each
    pixel; the box spans ten sigma unless ``boxsize`` says otherwise."""

    def __init__(self, sigma, integrate=True, boxsize=None):
        sigma = prepare_param(sigma, "sigma", fixed=True)
        super().__init__(sigma, integrate=integrate,
                         boxsize=int(np.ceil(10 * np.max(sigma))) if boxsize is None else boxsize)

    def get_model(self, *parameters, offset=None):
        return self._evaluate([(s,) for s in self.get_parameter(0, *parameters)], offset)

    def _profile(self, Y, X, sigma):
        return self._f(Y, sigma)[:, None] * self._f(X, sigma)[None, :]

    def _f(self, X, sigma):
        """1-D factor of the separable profile."""
        if not self.integrate:
            return np.exp(-(X**2) / (2 * sigma**2))
        # integral of the Gaussian over the pixel [X - 1/2, X + 1/2]
        a = np.sqrt(2) * sigma
        return np.sqrt(np.pi / 2) * sigma * (
            2 - special.erfc((0.5 - X) / a) - special.erfc((2 * X + 1) / (2 * a))
        )


class MoffatPSF(FunctionPSF):
    """Moffat profile ``(1 + r^2 / alpha^2)^-beta`` sampled at the pixel centres; the box
    spans five ``alpha`` unless ``boxsize`` says otherwise."""

    def __init__(self, alpha=4.7, beta=1.5, integrate=False, boxsize=None):
        assert integrate is False, "In-pixel integration not implemented (yet)!"
        alpha, beta = prepare_param(alpha, "alpha", fixed=True), prepare_param(beta, "beta", fixed=True)
        assert len(alpha) == len(beta)
        super().__init__(alpha, beta, integrate=integrate,
                         boxsize=int(np.ceil(5 * np.max(alpha))) if boxsize is None else boxsize)

    def get_model(self, *parameters, offset=None):
        alpha, beta = (self.get_parameter(i, *parameters) for i in (0, 1))
        return self._evaluate(list(zip(alpha, beta)), offset)

    def _profile(self, Y, X, a, b):
        return (1 + (X[None, :] ** 2 + Y[:, None] ** 2) / a**2) ** -b


class ImagePSF(PSF):
    """PSF from a centred image (2-D) or image cube (3-D), normalised per band."""

    def __init__(self, image):
        cube = image[np.newaxis] if image.ndim == 2 else image
        cube = prepare_param(normalize(cube), "image", fixed=True)
        super().__init__(cube)
        self.bbox = _centered_box(*cube.shape)

    def get_model(self, *parameters, offset=None):
        stored = self.get_parameter(0, *parameters)
        if offset is None:
            return stored.copy()
        return shift(stored.copy(), offset, return_Fourier=False)
