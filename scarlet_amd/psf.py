"""Point-spread-function models (reference scarlet/psf.py).  Set-up time only:
PSF images feed the difference-kernel construction in ``Observation.match``."""

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


class PSF(Model):
    @abstractmethod
    def get_model(self, *parameter, offset=None):
        """Centred PSF image cube, optionally shifted by ``offset`` pixels."""


class FunctionPSF(PSF):
    """PSF given by a profile function evaluated on an odd ``boxsize`` grid."""

    def __init__(self, *parameters, integrate=True, boxsize=None):
        super().__init__(*parameters)
        self.integrate = integrate
        if boxsize is None:
            boxsize = 15
        if boxsize % 2 == 0:
            boxsize += 1
        p0 = self.get_parameter(0, *parameters)
        half = boxsize // 2
        self.bbox = Box((len(p0), boxsize, boxsize), origin=(0, -half, -half))
        self._Y = np.arange(boxsize) - half
        self._X = np.arange(boxsize) - half
        self.is_same = np.all(p0 == p0[0])
        self._d = self.bbox.D - 2

    def expand_dims(self, model):
        return np.expand_dims(model, axis=tuple(range(self._d)))


class GaussianPSF(FunctionPSF):
    """Circular Gaussian, by default integrated over each pixel."""

    def __init__(self, sigma, integrate=True, boxsize=None):
        sigma = prepare_param(sigma, "sigma", fixed=True)
        if boxsize is None:
            boxsize = int(np.ceil(10 * np.max(sigma)))
        super().__init__(sigma, integrate=integrate, boxsize=boxsize)

    def get_model(self, *parameters, offset=None):
        sigma = self.get_parameter(0, *parameters)
        oy, ox = (0, 0) if offset is None else offset

        def image(s):
            return self._f(self._Y - oy, s)[:, None] * self._f(self._X - ox, s)[None, :]

        if self.is_same:
            psfs = self.expand_dims(image(sigma[0]))
        else:
            psfs = np.stack([image(s) for s in sigma], axis=0)
        return normalize(psfs)

    def _f(self, X, sigma):
        if not self.integrate:
            return np.exp(-(X**2) / (2 * sigma**2))
        # integral of the Gaussian over the pixel [X - 1/2, X + 1/2]
        a = np.sqrt(2) * sigma
        return np.sqrt(np.pi / 2) * sigma * (
            2 - special.erfc((0.5 - X) / a) - special.erfc((2 * X + 1) / (2 * a))
        )


class MoffatPSF(FunctionPSF):
    """Moffat profile ``(1 + r^2/alpha^2)^-beta`` (no pixel integration)."""

    def __init__(self, alpha=4.7, beta=1.5, integrate=False, boxsize=None):
        alpha = prepare_param(alpha, "alpha", fixed=True)
        beta = prepare_param(beta, "beta", fixed=True)
        assert len(alpha) == len(beta)
        assert integrate is False, "In-pixel integration not implemented (yet)!"
        if boxsize is None:
            boxsize = int(np.ceil(5 * np.max(alpha)))
        super().__init__(alpha, beta, integrate=integrate, boxsize=boxsize)

    def get_model(self, *parameters, offset=None):
        alpha = self.get_parameter(0, *parameters)
        beta = self.get_parameter(1, *parameters)
        oy, ox = (0, 0) if offset is None else offset
        Y, X = self._Y - oy, self._X - ox
        if self.is_same:
            psfs = self.expand_dims(self._f(Y, X, alpha[0], beta[0]))
        else:
            psfs = np.stack([self._f(Y, X, a, b) for a, b in zip(alpha, beta)], axis=0)
        return normalize(psfs)

    def _f(self, Y, X, a, b):
        return (1 + (X[None, :] ** 2 + Y[:, None] ** 2) / a**2) ** -b


class ImagePSF(PSF):
    """PSF from a centred image (2-D) or image cube (3-D), normalised per band."""

    def __init__(self, image):
        if image.ndim == 2:
            image = image.reshape(1, *image.shape)
        image = prepare_param(normalize(image), "image", fixed=True)
        super().__init__(image)
        self.bbox = Box(image.shape, origin=(0, -(image.shape[1] // 2), -(image.shape[2] // 2)))

    def get_model(self, *parameters, offset=None):
        image = self.get_parameter(0, *parameters).copy()
        if offset is not None:
            image = shift(image, offset, return_Fourier=False)
        return image
