"""Renderers map the model cube onto an observation (reference
scarlet/renderer.py:12-259).  Set-up (channel map, data box, difference kernel)
happens on the host; applying a renderer runs on the GPU."""

import numpy as np

from . import fft
from .bbox import Box, overlapped_slices
from .model import Model
from .parameter import Parameter


class Renderer(Model):
    def __init__(self, data_frame, model_frame, *parameters):
        self.data_frame = data_frame
        self.model_frame = model_frame
        self.channel_map = self.get_channel_map(data_frame, model_frame)
        super().__init__(*parameters)

    def __call__(self, model, *parameters):
        self.transform = self.get_model(*parameters)
        return self.transform(model)

    def get_channel_map(self, data_frame, model_frame):
        """``None`` for identical channel lists, a ``slice`` for a contiguous
        subset, otherwise the list of model-channel indices."""
        data_ch, model_ch = list(data_frame.channels), list(model_frame.channels)
        if data_ch == model_ch:
            return None
        idx = [model_ch.index(c) for c in data_ch]
        if max(idx) + 1 - min(idx) == len(idx):
            return slice(min(idx), max(idx) + 1)
        return idx

    def map_channels(self, model):
        if self.channel_map is None:
            return model
        return model[self.channel_map]


class NullRenderer(Renderer):
    """Observation and model share the PSF: rendering is the identity."""

    def __init__(self, data_frame, model_frame):
        super().__init__(data_frame, model_frame)

    def get_model(self, *parameters):
        return lambda model: model


class ConvolutionRenderer(Renderer):
    """Convolution with the difference kernel between the observed and the model
    PSF (``fft.match_psf(obs, model, padding)``, renderer.py:197-202)."""

    def __init__(self, data_frame, model_frame, *parameters, convolution_type="fft",
                 padding=10, psf_shift=None):
        if psf_shift is not None:
            # free sub-pixel shift of the difference kernel (renderer.py:175-177)
            psf_shift = Parameter(psf_shift, name="psf_shift", step=1.0e-2)
            parameters = (*parameters, psf_shift)
        super().__init__(data_frame, model_frame, *parameters)
        assert convolution_type in ["real", "fft"], "`convolution` must be either 'real' or 'fft'"
        self._convolution_type = convolution_type

        # region of the model frame covered by the data (translation only)
        pix = data_frame.convert_pixel_to(model_frame)
        lo = np.round(pix.min(axis=0)).astype("int")
        hi = np.round(pix.max(axis=0)).astype("int") + 1
        data_box = model_frame.bbox[0] @ Box.from_bounds((lo[0], hi[0]), (lo[1], hi[1]))
        self.slices = overlapped_slices(data_box, model_frame.bbox)

        dtype = model_frame.dtype
        self.diff_kernel = fft.match_psf(
            fft.Fourier(data_frame.psf.get_model().astype(dtype)),
            fft.Fourier(model_frame.psf.get_model().astype(dtype)),
            padding=padding,
        )

    def kernel_image(self, *parameters):
        """The difference kernel, moved by ``psf_shift`` if the renderer has one
        (renderer.py:215-228): Fourier shift of the stamp, cropped back to the stamp."""
        shift = self.get_parameter("psf_shift", *parameters)
        image = self.diff_kernel.image
        if shift is None:
            return image
        return fft.shift(image, shift, return_Fourier=False).astype(image.dtype)

    def kernel_derivatives(self, *parameters):
        """d kernel_image / d psf_shift[0], [1]."""
        shift = self.get_parameter("psf_shift", *parameters)
        return [d.astype(self.diff_kernel.image.dtype)
                for d in fft.shift_derivatives(self.diff_kernel.image, shift)]

    def get_model(self, *parameters):
        def transform(model, *parameters):
            from .observation import _device_render

            return _device_render(self, model, self.kernel_image(*parameters))

        return transform

    def __call__(self, model, *parameters):
        self.transform = self.get_model(*parameters)
        return self.transform(model, *parameters)
