"""Host-side Fourier utilities with the reference's ``scarlet.fft`` interface.

These run once per ``Observation.match`` (difference-kernel construction) and
in scene set-up; the per-iteration convolution of the fitting loop never comes
here -- it runs on the GPU (``csrc/fftconv.hip``).  Names, argument meaning and
centring conventions are those of the reference (scarlet/fft.py): an odd-sized
array is centred on the centre-right pixel of an even-sized one, both when
padding (``_pad``, fft.py:82-113) and when cropping (``_centered``, fft.py:9-36).
"""

import operator

import numpy as np


def _is_5smooth(n):
    for p in (2, 3, 5):
        while n % p == 0:
            n //= p
    return n == 1


def next_fast_len(n):
    """Next length >= n whose only prime factors are 2, 3, 5 (the sizes the
    reference obtains from ``scipy.fftpack.next_fast_len``, fft.py:155)."""
    n = max(int(n), 1)
    while not _is_5smooth(n):
        n += 1
    return n


def _axis_list(axes, ndim):
    if axes is None:
        return list(range(ndim))
    try:
        return list(axes)
    except TypeError:
        return [axes]


def _centered(arr, newshape):
    """Central ``newshape`` region of ``arr`` (start index (cur-new+1)//2)."""
    newshape = np.asarray(newshape)
    cur = np.array(arr.shape)
    if not np.all(newshape <= cur):
        raise ValueError(
            "arr must be larger than newshape in both dimensions, received "
            "{0}, and {1}".format(arr.shape, newshape)
        )
    lo = (cur - newshape + 1) // 2
    return arr[tuple(slice(a, a + n) for a, n in zip(lo, newshape))]


def fast_zero_pad(arr, pad_width):
    """Zero padding by allocation + slice assignment (fft.py:39-67)."""
    shape = [n + a + b for n, (a, b) in zip(arr.shape, pad_width)]
    out = np.zeros(shape, dtype=arr.dtype)
    out[tuple(slice(a, a + n) for n, (a, _) in zip(arr.shape, pad_width))] = arr
    return out


def _pad(arr, newshape, axes=None, mode="constant", constant_values=0):
    """Pad ``arr`` to ``newshape`` along ``axes`` keeping the fftshift centre."""
    ax = _axis_list(axes, arr.ndim)
    widths = [(0, 0)] * arr.ndim
    for a, axis in enumerate(ax):
        extra = int(newshape[a]) - arr.shape[axis]
        before = (extra + 1) // 2
        widths[axis] = (before, extra - before)
    if mode == "constant" and constant_values == 0:
        return fast_zero_pad(arr, widths)
    return np.pad(arr, widths, mode=mode)


def _get_fft_shape(im_or_shape1, im_or_shape2, padding=3, axes=None, max=False):
    """Fast FFT shape for combining two images along ``axes`` (fft.py:116-167)."""
    s1 = np.asarray(getattr(im_or_shape1, "shape", im_or_shape1))
    s2 = np.asarray(getattr(im_or_shape2, "shape", im_or_shape2))
    if len(s1) != len(s2):
        raise ValueError(
            "img1 and img2 must have the same number of dimensions, but got "
            "{0} and {1}".format(len(s1), len(s2))
        )
    ax = _axis_list(axes, len(s1))
    if max:
        sizes = [int(np.maximum(s1[a], s2[a])) for a in ax]
    else:
        sizes = [int(s1[a] + s2[a]) for a in ax]
    shape = [next_fast_len(s + padding) for s in sizes]
    # real transforms want an even last axis; an even-height kernel also needs
    # an even second-to-last axis so that its centre maps onto index 0
    while shape[-1] % 2:
        shape[-1] = next_fast_len(shape[-1] + 1)
    if s2[-2] % 2 == 0:
        while shape[-2] % 2:
            shape[-2] = next_fast_len(shape[-2] + 1)
    return shape


class Fourier:
    """An image together with its cached transforms, keyed by
    (fft_shape, axes, all_axes) as in the reference (fft.py:170-313)."""

    def __init__(self, image, image_fft=None):
        self._image = image
        self._fft = {} if image_fft is None else image_fft

    @staticmethod
    def from_fft(image_fft, fft_shape, image_shape, axes=None):
        if axes is None:
            axes = range(len(image_fft))
        axes = tuple(axes)
        img = np.fft.irfftn(image_fft, fft_shape, axes=axes)
        img = _centered(np.fft.fftshift(img, axes=axes), image_shape)
        key = (tuple(fft_shape), axes, tuple(range(len(image_shape))))
        return Fourier(img, {key: image_fft})

    @property
    def image(self):
        return self._image

    @property
    def shape(self):
        return self._image.shape

    def fft(self, fft_shape, axes):
        axes = tuple(_axis_list(axes, self._image.ndim))
        key = (tuple(fft_shape), axes, tuple(range(self._image.ndim)))
        if key not in self._fft:
            if len(fft_shape) != len(axes):
                raise ValueError(
                    "fft_shape self.axes must have the same number of dimensions, "
                    "got {0}, {1}".format(fft_shape, axes)
                )
            padded = _pad(self._image, fft_shape, axes)
            self._fft[key] = np.fft.rfftn(np.fft.ifftshift(padded, axes), axes=axes)
        return self._fft[key]

    def __len__(self):
        return len(self._image)

    def __getitem__(self, index):
        if not hasattr(index, "__getitem__"):
            index = (index,)
        dropped = [
            n for n, i in enumerate(index) if not isinstance(i, slice) and i is not None
        ]
        ffts = {}
        for (shape, axes, all_axes), value in self._fft.items():
            keep = [k for k, a in enumerate(axes) if a not in dropped]
            key = (
                tuple(shape[k] for k in keep),
                tuple(axes[k] for k in keep),
                tuple(a for a in all_axes if a not in dropped),
            )
            ffts[key] = value[index]
        return Fourier(self._image[index], ffts)


def _kspace_operation(image1, image2, padding, op, shape, axes):
    if len(image1.shape) != len(image2.shape):
        raise Exception(
            "Both images must have the same number of axes, got {0} and {1}".format(
                len(image1.shape), len(image2.shape)
            )
        )
    fft_shape = _get_fft_shape(image1.image, image2.image, padding, axes)
    spectrum = op(image1.fft(fft_shape, axes), image2.fft(fft_shape, axes))
    return Fourier.from_fft(spectrum, fft_shape, shape, axes)


def _as_fourier(x):
    return x if isinstance(x, Fourier) else Fourier(x)


def match_psf(psf1, psf2, padding=3, axes=(-2, -1), return_Fourier=True):
    """Difference kernel that maps ``psf2`` onto ``psf1`` (k-space ratio)."""
    psf1, psf2 = _as_fourier(psf1), _as_fourier(psf2)
    shape = psf2.shape if psf1.shape[0] < psf2.shape[0] else psf1.shape
    diff = _kspace_operation(psf1, psf2, padding, operator.truediv, shape, axes=axes)
    return diff if return_Fourier else np.real(diff.image)


def convolve(image, kernel, padding=3, axes=(-2, -1), return_Fourier=True):
    """Zero-boundary convolution of ``image`` with ``kernel`` ('same' size)."""
    image, kernel = _as_fourier(image), _as_fourier(kernel)
    out = _kspace_operation(image, kernel, padding, operator.mul, image.shape, axes=axes)
    return out if return_Fourier else np.real(out.image)


def mk_shifter(shape, real=False):
    """Per-axis Fourier phase ramps ``-2 pi i f`` (interpolation.py:341-375)."""
    fy = np.fft.rfftfreq(shape[-2]) if real else np.fft.fftfreq(shape[-2])
    fx = np.fft.rfftfreq(shape[-1])
    return -2j * np.pi * fy, -2j * np.pi * fx


def shift(image, shift, fft_shape=None, axes=(-2, -1), return_Fourier=True):
    """Sub-pixel translation through a Fourier phase ramp (fft.py:399-428)."""
    if fft_shape is None:
        fft_shape = _get_fft_shape(image, image, padding=10, axes=axes)
    ramp_y, ramp_x = mk_shifter(fft_shape)
    image = _as_fourier(image)
    spectrum = image.fft(fft_shape, axes)
    ramp = np.exp(ramp_y[:, None] * shift[0]) * np.exp(ramp_x[None, :] * shift[1])
    nd = len(image.shape)
    if nd > 2:
        lead = tuple(d for d in range(nd) if d not in axes and d - nd not in axes)
        ramp = np.expand_dims(ramp, axis=lead)
    out = Fourier.from_fft(spectrum * ramp, fft_shape, image.shape, axes)
    return out if return_Fourier else np.real(out.image)


def shift_derivatives(image, shift, fft_shape=None, axes=(-2, -1)):
    """``(d/d shift[0], d/d shift[1])`` of ``shift(image, shift, return_Fourier=False)``:
    the same pipeline with the spectrum multiplied by the ramp of the differentiated axis
    (what autograd makes of fft.py:399-428)."""
    if fft_shape is None:
        fft_shape = _get_fft_shape(image, image, padding=10, axes=axes)
    ramp_y, ramp_x = mk_shifter(fft_shape)
    image = _as_fourier(image)
    spectrum = image.fft(fft_shape, axes)
    nd = len(image.shape)
    lead = tuple(d for d in range(nd) if d not in axes and d - nd not in axes)

    def expand(a):
        return np.expand_dims(a, axis=lead) if nd > 2 else a

    ramp = expand(np.exp(ramp_y[:, None] * shift[0]) * np.exp(ramp_x[None, :] * shift[1]))
    out = []
    for factor in (expand(ramp_y[:, None] * np.ones_like(ramp_x)[None, :]),
                   expand(np.ones_like(ramp_y)[:, None] * ramp_x[None, :])):
        out.append(np.real(Fourier.from_fft(spectrum * ramp * factor, fft_shape, image.shape,
                                            axes).image))
    return out
