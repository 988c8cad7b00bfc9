"""Observations: data, weights and the renderer that maps the model onto them
(reference scarlet/observation.py)."""

import numpy as np
import numpy.ma as ma

from .bbox import overlapped_slices
from .frame import Frame
from .renderer import ConvolutionRenderer, NullRenderer, Renderer


def _device_render(renderer, model, kernel=None):
    """``model`` (C, H, W) convolved with the renderer's difference kernel on
    the GPU: the cube is presented to the batch as C unit-spectrum components."""
    from .batch import BlendBatch, ComponentSpec

    model_ = np.ascontiguousarray(renderer.map_channels(model), dtype=np.float32)
    C, H, W = model_.shape
    eye = np.eye(C, dtype=np.float32)
    comps = [ComponentSpec(eye[c], model_[c], (0, 0), prox_flags=0) for c in range(C)]
    zeros = np.zeros((1, C, H, W), dtype=np.float32)
    kernel = np.ascontiguousarray(
        renderer.diff_kernel.image if kernel is None else kernel, dtype=np.float32)
    batch = BlendBatch(zeros, zeros + 1, [comps], kernel=kernel, max_iter=1)
    try:
        _, rendered, _ = batch.forward(model=False)
    finally:
        batch.close()
    out = rendered[0].astype(renderer.data_frame.dtype, copy=False)
    data_sl, model_sl = renderer.slices
    if out[model_sl].shape == tuple(renderer.data_frame.shape):
        return out[model_sl]
    matched = np.zeros(renderer.data_frame.shape, dtype=renderer.data_frame.dtype)
    matched[data_sl] = out[model_sl]
    return matched


class Observation(Frame):
    """Data cube (channels, Ny, Nx) with inverse-variance ``weights`` (zero for
    masked pixels), ``psf`` and ``channels``."""

    def __init__(self, data, channels, psf=None, weights=None, wcs=None, padding=10):
        super().__init__(data.shape, wcs=wcs, psf=psf, channels=channels, dtype=data.dtype)
        self.data = data
        self.weights = np.ones(data.shape, dtype=data.dtype) if weights is None else weights
        assert self.weights.shape == self.data.shape, "Weights needs to have same shape as data"
        self._padding = padding

    def match(self, model_frame, renderer=None):
        """Bind this observation to ``model_frame``: data and (array) weights take the
        model's dtype, and ``self.renderer`` becomes ``renderer`` if one is given or the
        default for the pair of frames (reference observation.py:58-113).  Returns
        ``self`` so that ``Observation(...).match(frame)`` chains."""
        if renderer is not None:
            assert isinstance(renderer, Renderer)
        self.model_frame = model_frame
        self._cast_to(model_frame.dtype)
        self.renderer = self._default_renderer(model_frame) if renderer is None else renderer
        return self

    def _cast_to(self, dtype):
        if self.dtype == dtype:
            return
        self.dtype = dtype
        self.data = self.data.astype(dtype)
        # weights may be a masked array or another array-like; only plain arrays are cast
        if type(self.weights) is np.ndarray:
            self.weights = self.weights.astype(dtype)

    def _default_renderer(self, model_frame):
        """``NullRenderer`` for a shared PSF object; FFT convolution with the difference
        kernel when the pixel grids agree (same WCS object, or WCSs that differ by a
        translation only); ``ResolutionRenderer`` when scale or orientation differ."""
        if self.psf is model_frame.psf:
            return NullRenderer(self, model_frame)
        assert self.psf is not None and model_frame.psf is not None
        if self.wcs is not model_frame.wcs:
            from . import interpolation
            from .renderer import ResolutionRenderer

            assert self.wcs is not None and model_frame.wcs is not None
            (_, sin_rot), scale = interpolation.get_angles(self.wcs, model_frame.wcs)
            tiny = np.finfo(float).eps
            if abs(scale - 1) >= tiny or abs(sin_rot) ** 2 >= tiny:
                return ResolutionRenderer(self, model_frame)
        return ConvolutionRenderer(self, model_frame, convolution_type="fft")

    @property
    def noise_rms(self):
        """Per-pixel noise rms ``weights ** -1/2`` as a masked array: pixels of zero
        weight are masked and read as infinite when filled."""
        try:
            return self._noise_rms
        except AttributeError:
            rms = 1 / np.sqrt(ma.masked_equal(self.weights, 0))
            rms.fill_value = np.inf
            self._noise_rms = rms
            return rms

    @property
    def parameters(self):
        return self.renderer.parameters

    def render(self, model, *parameters):
        """Model cube mapped into the observation frame (on the GPU)."""
        return self.renderer(model, *parameters)

    def get_log_likelihood(self, model, *parameters, noise_factor=0):
        """``-log_norm - sum w (render(model) - data)^2 / 2``."""
        model_ = self.render(model, *parameters)
        data_, weights_ = self.noisy(noise_factor) if noise_factor > 0 else (self.data, self.weights)
        return -self.log_norm - np.sum(weights_ * (model_ - data_) ** 2) / 2

    def noisy(self, noise_factor):
        """``(data, weights)`` of one evaluation with ``noise_factor > 0``
        (observation.py:165-168): a fresh draw of the pixel noise from NumPy's global
        generator is added to the data, the weights are divided by ``noise_factor + 1``."""
        noise = np.random.normal(loc=0, scale=self.noise_rms)
        # pixels without weight have no noise level; they do not enter the likelihood
        noise = np.where(ma.getmaskarray(self.noise_rms), 0, np.asarray(noise))
        return self.data + noise.astype(self.data.dtype), self.weights / (noise_factor + 1)

    @property
    def log_norm(self):
        if not hasattr(self, "_log_norm"):
            n = np.prod(self.data.shape) - self.noise_rms.mask.sum()
            with np.errstate(divide="ignore"):
                self._log_norm = n / 2 * np.log(2 * np.pi) + np.log(self.noise_rms).sum()
        return self._log_norm

    def _to_frame(self, frame, data=None):
        frame_sl, obs_sl = overlapped_slices(frame.bbox, self.bbox)
        if data is None:
            data = self.data
        out = np.zeros(frame.shape, dtype=getattr(frame, "dtype", data.dtype))
        out[frame_sl] = data[obs_sl]
        return out
