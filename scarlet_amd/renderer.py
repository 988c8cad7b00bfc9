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

    def _overlap_slices(self):
        """(data slices, model slices) of the region where an observation on the model's
        pixel grid (translation only) and the model frame overlap: what ``match_shape``
        crops / zero-embeds (renderer.py:130-161, 184-195)."""
        pix = self.data_frame.convert_pixel_to(self.model_frame)
        lo = np.round(pix.min(axis=0)).astype("int")
        hi = np.round(pix.max(axis=0)).astype("int") + 1
        data_box = self.model_frame.bbox[0] @ Box.from_bounds((lo[0], hi[0]), (lo[1], hi[1]))
        return overlapped_slices(data_box, self.model_frame.bbox)


class NullRenderer(Renderer):
    """Observation and model share the PSF: rendering is the identity."""

    def __init__(self, data_frame, model_frame):
        super().__init__(data_frame, model_frame)
        if tuple(data_frame.shape[1:]) == tuple(model_frame.shape[1:]):
            full = (slice(None),) * 3
            self.slices = (full, full)
        else:
            self.slices = self._overlap_slices()

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
        self.slices = self._overlap_slices()

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

    def render_float64(self, model, *parameters):
        """The same mapping evaluated in double precision on the host, like the reference's
        NumPy path does when it is handed a float64 model (renderer.py:215-259 ->
        fft.convolve): for set-up computations that amplify rounding, i.e. the normal
        equations of ``initialization.set_spectra_to_match`` (condition numbers of several
        hundred).  Not used by the fitting loop."""
        model_ = np.asarray(self.map_channels(model), dtype=np.float64)
        if self.get_parameter("psf_shift", *parameters) is not None:
            kernel = fft.Fourier(np.asarray(self.kernel_image(*parameters), dtype=np.float64))
        else:
            # (the kernel's transforms are kept with it: one per FFT shape, not one per call)
            kernel = getattr(self, "_kernel_float64", None)
            if kernel is None:
                kernel = self._kernel_float64 = fft.Fourier(
                    np.asarray(self.kernel_image(), dtype=np.float64))
        # A component's model is zero outside its box: convolve the part of the frame the
        # kernel can reach from there (zero-boundary convolution: the same sums, on a
        # transform a fraction of the frame's size) and leave the rest zero.
        out = None
        rows = np.flatnonzero(model_.any(axis=(0, 2)))
        cols = np.flatnonzero(model_.any(axis=(0, 1)))
        if rows.size == 0:
            out = np.zeros(model_.shape, dtype=np.float64)
        else:
            hy, hx = kernel.shape[-2] // 2 + 1, kernel.shape[-1] // 2 + 1
            y0, y1 = max(rows[0] - hy, 0), min(rows[-1] + 1 + hy, model_.shape[1])
            x0, x1 = max(cols[0] - hx, 0), min(cols[-1] + 1 + hx, model_.shape[2])
            if 2 * (y1 - y0) * (x1 - x0) <= model_.shape[1] * model_.shape[2]:
                out = np.zeros(model_.shape, dtype=np.float64)
                cut = np.ascontiguousarray(model_[:, y0:y1, x0:x1])
                out[:, y0:y1, x0:x1] = fft.convolve(fft.Fourier(cut), kernel, axes=(1, 2)).image
        if out is None:
            out = fft.convolve(fft.Fourier(model_), kernel, axes=(1, 2)).image
        data_sl, model_sl = self.slices
        if out[model_sl].shape == tuple(self.data_frame.shape):
            return out[model_sl]
        matched = np.zeros(self.data_frame.shape, dtype=np.float64)
        matched[data_sl] = out[model_sl]
        return matched


class ResolutionRenderer(Renderer):
    """Renders a high-resolution model into a low-resolution observation with a
    different pixel scale (reference renderer.py:262-547): convolution with the
    difference kernel between the observed PSF and the model PSF, and resampling onto
    the observation's pixels, in one linear operator.

    Set-up on the host as in the reference: the observed PSF is sinc-interpolated to the
    model pixel scale, the difference kernel is padded to the FFT shape and Fourier-
    shifted along y to every low-resolution row (``_resconv_op``).  The per-call part --
    shifting the padded model along x to every low-resolution column and contracting
    with the operator (renderer.py:478-545) -- is linear in the model; it runs on the
    GPU (``smi_resampler_*``), through transforms along x since the shift operator is
    circulant, or as two dense products per band (``device_path``).  Rotated grids are not
    supported."""

    def __init__(self, data_frame, model_frame, padding=10):
        from . import interpolation

        super().__init__(data_frame, model_frame)
        self.angle, self.h = interpolation.get_angles(data_frame.wcs, model_frame.wcs)
        self.isrot = (np.abs(self.angle[1]) ** 2) > np.finfo(float).eps
        if self.isrot:
            raise NotImplementedError("ResolutionRenderer between rotated pixel grids")
        lr_shape = data_frame.shape[1:]
        pixels = np.stack((np.arange(lr_shape[0]), np.arange(lr_shape[1])), axis=1)
        coord_hr = data_frame.convert_pixel_to(model_frame, pixel=pixels)
        diff_psf, psf_lr_hr = self.build_diffkernel(data_frame, model_frame)
        # (np.stack above, renderer.py:274 in the reference, takes the two pixel ranges as
        # columns of one array: it raises for every observation that is not square -- checked
        # by running the reference on 28 x 38 and 38 x 28 crops, DESIGN.md 8.5 -- and a
        # square one has small_axis = True, so the reference's other unrotated branch
        # (renderer.py:354-363, 536-545) cannot be reached and has no counterpart here)
        self.small_axis = data_frame.Nx <= data_frame.Ny
        assert self.small_axis
        self._fft_shape = fft._get_fft_shape(psf_lr_hr, np.zeros(model_frame.shape), padding=3,
                                             axes=[-2, -1], max=False)
        if (self._fft_shape[-2] < diff_psf.shape[-2]) or (self._fft_shape[-1] < diff_psf.shape[-1]):
            diff_psf = fft.Fourier(fft._centered(
                diff_psf.image, np.array([diff_psf.shape[0] + 1, *self._fft_shape]) - 1))
        self.diff_kernel = fft.Fourier(fft._pad(diff_psf.image, self._fft_shape, axes=(-2, -1)))
        Fy, Fx = self._fft_shape
        center_y = int(Fy / 2.0 - (Fy - model_frame.Ny) / 2.0) + ((Fy % 2) != 0) * (
            (model_frame.Ny % 2) == 0)
        center_x = int(Fx / 2.0 - (Fx - model_frame.Nx) / 2.0) - ((Fx % 2) != 0) * (
            (model_frame.Nx % 2) == 0)
        self.shifts = coord_hr.T.copy()
        self.shifts[0] -= center_y
        self.shifts[1] -= center_x
        self.other_shifts = np.copy(self.shifts)
        # the kernel shifted along y to every low-resolution row (renderer.py:341-353)
        op = self._shift_along(self.diff_kernel.image, self.shifts[0], axis=1)  # (C, n_a, Fy, Fx)
        self._resconv_op = (np.array(op, dtype=model_frame.dtype) * self.h**2).reshape(
            op.shape[0], op.shape[1], -1)
        self._device = None

    def build_diffkernel(self, data_frame, model_frame):
        """Difference kernel between the observed PSF, interpolated to the model pixels,
        and the model PSF (renderer.py:365-412)."""
        from . import interpolation

        psf_hr = model_frame.psf.get_model()
        psf_lr = data_frame.psf.get_model().astype(model_frame.dtype)
        pad_shape = np.array((np.array(data_frame.shape[-2:]) + np.array(psf_lr.shape[-2:])) / 2
                             ).astype(int) * 2 + 1
        h_lr = interpolation.get_pixel_size(interpolation.get_affine(data_frame.wcs))
        h_hr = interpolation.get_pixel_size(interpolation.get_affine(model_frame.wcs))
        angle, _ = interpolation.get_angles(model_frame.wcs, data_frame.wcs)
        psf_lr_hr = interpolation.sinc_interp_inplace(psf_lr, h_lr, h_hr, angle,
                                                      pad_shape=pad_shape)
        psf_hr = psf_hr / np.sum(psf_hr)
        psf_lr_hr = psf_lr_hr / np.sum(psf_lr_hr)
        return fft.match_psf(fft.Fourier(psf_lr_hr), fft.Fourier(psf_hr)), psf_hr

    def _shift_along(self, cube, shifts, axis):
        """``cube`` (C, Fy, Fx) Fourier-shifted along ``axis`` (1 = y, 2 = x) by every
        entry of ``shifts`` (renderer.py:414-476, one-axis branches: real transform along
        that axis, ramps of ``mk_shifter(real=True)``).  Returns (C, n, Fy, Fx) for
        axis 1 and (C, Fy, Fx, n) for axis 2."""
        from .interpolation import mk_shifter

        F = self._fft_shape[axis - 1]
        ramp = mk_shifter(self._fft_shape, real=True)[axis - 1]
        spectrum = np.fft.rfft(np.fft.ifftshift(cube, axes=axis), axis=axis)
        if axis == 1:
            phase = np.exp(ramp[np.newaxis, :] * shifts[:, np.newaxis])  # (n, Fy/2+1)
            shifted = spectrum[:, np.newaxis, :, :] * phase[np.newaxis, :, :, np.newaxis]
            return np.fft.fftshift(np.fft.irfft(shifted, F, axis=2), axes=2)
        phase = np.exp(ramp[:, np.newaxis] * shifts[np.newaxis, :])  # (Fx/2+1, n)
        shifted = spectrum[:, :, :, np.newaxis] * phase[np.newaxis, np.newaxis, :, :]
        return np.fft.fftshift(np.fft.irfft(shifted, F, axis=2), axes=2)

    def _resampler(self):
        """Device handle holding the two operators (built on first use)."""
        if self._device is None:
            import ctypes

            from . import _lib

            lib = _lib.load()
            C, n_a, _ = self._resconv_op.shape
            Fy, Fx = self._fft_shape
            n_b = self.other_shifts.shape[1]
            # operator that shifts one padded row to every low-resolution column: the
            # reference's pipeline applied to the identity (row x' -> P[x, b, x'])
            eye = np.eye(Fx, dtype=np.float64)[np.newaxis]  # (1, Fx rows, Fx)
            P = self._shift_along(eye, -self.other_shifts[1], axis=2)[0]  # (x', x, b)
            Pt = np.ascontiguousarray(P.reshape(Fx, Fx * n_b), dtype=np.float32)
            A = np.ascontiguousarray(self._resconv_op, dtype=np.float32)
            handle = ctypes.c_void_p()
            _lib.check(lib.smi_resampler_create(
                _lib.ptr(A, ctypes.c_float), _lib.ptr(Pt, ctypes.c_float), C, n_a, n_b, Fy, Fx,
                ctypes.byref(handle)))
            self._device = (lib, handle, (C, n_a, n_b))
        return self._device

    def device_path(self, path=None):
        """How the device evaluates the operator: 1 = through transforms along x (the shift
        operator the reference builds is circulant), 0 = two dense products per band.
        ``device_path(0)`` switches to the dense products (parity checks)."""
        import ctypes

        from . import _lib

        lib, handle, _ = self._resampler()
        if path is not None:
            _lib.check(lib.smi_resampler_set_path(handle, int(path)))
        now = ctypes.c_int32(-1)
        _lib.check(lib.smi_resampler_get_path(handle, ctypes.byref(now)))
        return now.value

    def get_model(self, *parameters):
        def transform(model, *parameters):
            import ctypes

            from . import _lib

            model_ = self.map_channels(model)
            dtype = model_.dtype
            padded = np.ascontiguousarray(fft._pad(model_, self._fft_shape, axes=(-2, -1)),
                                          dtype=np.float32)
            lib, handle, shape = self._resampler()
            out = np.empty(shape, dtype=np.float32)
            _lib.check(lib.smi_resampler_render(handle, _lib.ptr(padded, ctypes.c_float),
                                                _lib.ptr(out, ctypes.c_float)))
            return out.astype(dtype, copy=False)

        return transform

    def __call__(self, model, *parameters):
        return self.get_model(*parameters)(model, *parameters)

    def __del__(self):
        try:
            if self._device is not None:
                self._device[0].smi_resampler_destroy(self._device[1])
        except Exception:
            pass
