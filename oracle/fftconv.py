"""Oracle: centring / padding / FFT-convolution arithmetic of ``scarlet/fft.py``.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  NumPy restatement; the
reference lines each function follows are cited.  On NumPy >= 2 ``rfftn`` of a
float32 array is complex64, exactly what the reference computes in the same
interpreter.
"""

import numpy as np


def next_fast_len(n):
    """Smallest 5-smooth integer >= n.

    ``scipy.fftpack.next_fast_len`` (called at scarlet/fft.py:155,160,164)
    returns the next size of the form 2^a 3^b 5^c.
    """
    n = int(n)
    if n <= 1:
        return 1
    best = None
    p5 = 1
    while p5 < 2 * n:
        p35 = p5
        while p35 < 2 * n:
            v = p35
            while v < n:
                v *= 2
            if best is None or v < best:
                best = v
            p35 *= 3
        p5 *= 5
    return best


def centered(arr, newshape):
    """Central ``newshape`` part of ``arr``; start = (cur-new+1)//2 (fft.py:9-36)."""
    newshape = tuple(int(s) for s in newshape)
    cur = arr.shape
    if any(n > c for n, c in zip(newshape, cur)):
        raise ValueError(
            "arr must be larger than newshape in both dimensions, received "
            "{0}, and {1}".format(arr.shape, newshape)
        )
    sl = []
    for c, n in zip(cur, newshape):
        start = (c - n + 1) // 2
        sl.append(slice(start, start + n))
    return arr[tuple(sl)]


def zero_pad(arr, pad_width):
    """``fast_zero_pad`` (fft.py:39-67): embed ``arr`` in zeros."""
    shape = tuple(a + lo + hi for a, (lo, hi) in zip(arr.shape, pad_width))
    out = np.zeros(shape, dtype=arr.dtype)
    sl = tuple(slice(lo, s - hi) for s, (lo, hi) in zip(shape, pad_width))
    out[sl] = arr
    return out


def pad_to(arr, newshape, axes=None):
    """``_pad`` (fft.py:82-113): zero-pad so that an odd-sized centre lands on
    the centre-right pixel of an even target: before = (dS+1)//2."""
    if axes is None:
        axes = tuple(range(arr.ndim))
        newshape = tuple(newshape)
    else:
        try:
            axes = tuple(axes)
        except TypeError:
            axes = (axes,)
    pw = [(0, 0)] * arr.ndim
    for a, ax in enumerate(axes):
        d = int(newshape[a]) - arr.shape[ax]
        lo = (d + 1) // 2
        pw[ax] = (lo, d - lo)
    return zero_pad(arr, pw)


def fft_shape(shape1, shape2, padding=3, axes=None):
    """``_get_fft_shape`` (fft.py:116-167), ``max=False`` branch.

    Per transformed axis: s1+s2+padding -> next 5-smooth length; the last axis
    is forced even; the second-to-last is forced even when the *second*
    operand (the kernel) has even height.
    """
    shape1 = tuple(getattr(shape1, "shape", shape1))
    shape2 = tuple(getattr(shape2, "shape", shape2))
    if len(shape1) != len(shape2):
        raise ValueError(
            "img1 and img2 must have the same number of dimensions, but got "
            "{0} and {1}".format(len(shape1), len(shape2))
        )
    if axes is None:
        axes = tuple(range(len(shape1)))
    else:
        try:
            axes = tuple(axes)
        except TypeError:
            axes = (axes,)
    shape = [next_fast_len(shape1[ax] + shape2[ax] + padding) for ax in axes]
    while shape[-1] % 2 != 0:
        shape[-1] = next_fast_len(shape[-1] + 1)
    if shape2[-2] % 2 == 0:
        while shape[-2] % 2 != 0:
            shape[-2] = next_fast_len(shape[-2] + 1)
    return shape


def forward_fft(image, shape, axes):
    """``Fourier.fft`` (fft.py:255-273): pad -> ifftshift -> rfftn."""
    padded = pad_to(image, shape, axes)
    return np.fft.rfftn(np.fft.ifftshift(padded, axes), axes=axes)


def inverse_fft(image_fft, shape, image_shape, axes):
    """``Fourier.from_fft`` (fft.py:200-243): irfftn -> fftshift -> centred crop."""
    img = np.fft.irfftn(image_fft, shape, axes=axes)
    img = np.fft.fftshift(img, axes=axes)
    return centered(img, image_shape)


def _kspace(image1, image2, padding, op, out_shape, axes):
    """``_kspace_operation`` (fft.py:316-331)."""
    if image1.ndim != image2.ndim:
        raise Exception(
            "Both images must have the same number of axes, got {0} and {1}".format(
                image1.ndim, image2.ndim
            )
        )
    shape = fft_shape(image1.shape, image2.shape, padding, axes)
    prod = op(forward_fft(image1, shape, axes), forward_fft(image2, shape, axes))
    return inverse_fft(prod, shape, out_shape, axes)


def match_psf(psf1, psf2, padding=3, axes=(-2, -1)):
    """Difference kernel ``psf1 (/) psf2`` in k-space (fft.py:334-365).

    The output stamp is the one of the operand with more channels
    (comparison of ``shape[0]``, fft.py:356-359).
    """
    out_shape = psf2.shape if psf1.shape[0] < psf2.shape[0] else psf1.shape
    return _kspace(psf1, psf2, padding, np.divide, out_shape, axes)


def convolve(image, kernel, padding=3, axes=(-2, -1)):
    """``fft.convolve`` (fft.py:368-396): zero-boundary 'same' convolution."""
    return _kspace(image, kernel, padding, np.multiply, image.shape, axes)


def convolve_adjoint(image, kernel, padding=3, axes=(-2, -1)):
    """Transpose of :func:`convolve` w.r.t. ``image``.

    Transposing pad/ifftshift/rfftn/x/irfftn/fftshift/crop term by term gives
    the same pipeline with the conjugated kernel spectrum; this is what the
    autograd vjp of fft.py:316-331 evaluates and what the reference authors
    write as a convolution with ``K[:, ::-1, ::-1]`` (lite/models.py:363-367).
    """
    shape = fft_shape(image.shape, kernel.shape, padding, axes)
    prod = forward_fft(image, shape, axes) * np.conj(forward_fft(kernel, shape, axes))
    return inverse_fft(prod, shape, image.shape, axes)


def filter_bounds(kernel2d):
    """Tap list for ``apply_filter`` (interpolation.py:7-65): per tap (cy, cx)
    relative to the stamp centre, start = max(0, c), end = -min(0, c)."""
    h, w = kernel2d.shape
    if h % 2 == 0 or w % 2 == 0:
        raise ValueError("ambiguous centre: the stamp must have odd height and width")
    cy, cx = np.meshgrid(np.arange(h) - h // 2, np.arange(w) - w // 2, indexing="ij")
    cy = cy.reshape(-1)
    cx = cx.reshape(-1)
    z = np.zeros_like(cy)
    return (
        np.maximum(z, cy).astype(np.int32),
        (-np.minimum(z, cy)).astype(np.int32),
        np.maximum(z, cx).astype(np.int32),
        (-np.minimum(z, cx)).astype(np.int32),
    )


def apply_filter(image, kernel2d):
    """Real-space 'same' convolution, the arithmetic of the reference's native
    ``apply_filter`` (operators_pybind11.cc:39-56): for every tap n,
    ``result[ys:, xs:] += v[n] * image[ye:, xe:]`` over the overlapping block."""
    ys, ye, xs, xe = filter_bounds(kernel2d)
    vals = kernel2d.reshape(-1)
    H, W = image.shape
    out = np.zeros_like(image)
    for n in range(vals.size):
        rows = H - ys[n] - ye[n]
        cols = W - xs[n] - xe[n]
        if rows <= 0 or cols <= 0:
            continue
        out[ys[n] : ys[n] + rows, xs[n] : xs[n] + cols] += (
            vals[n] * image[ye[n] : ye[n] + rows, xe[n] : xe[n] + cols]
        )
    return out


def fourier_shift(image, shift, axes=(-2, -1)):
    """Sub-pixel shift by a Fourier phase ramp (fft.py:399-428 with
    interpolation.py:341-375): padding 10, ``image (+) image`` fast shape."""
    shape = fft_shape(image.shape, image.shape, padding=10, axes=axes)
    fy = -2j * np.pi * np.fft.fftfreq(shape[0])
    fx = -2j * np.pi * np.fft.rfftfreq(shape[1])
    ramp = np.exp(fy[:, None] * shift[0]) * np.exp(fx[None, :] * shift[1])
    if image.ndim > 2:
        ramp = ramp.reshape((1,) * (image.ndim - 2) + ramp.shape)
    return inverse_fft(forward_fft(image, shape, axes) * ramp, shape, image.shape, axes)


class ShiftOperator:
    """:func:`fourier_shift` of an (h, w) image written out as the linear map it is:

        shifted = Dr @ x @ Tx.T - Di @ x @ Hx.T

    with Toeplitz matrices ``M[n, n'] = v(n - n')`` built from the phase ramps of
    fft.py:399-428 / interpolation.py:341-375 (``fftfreq`` along y, ``rfftfreq`` along x
    whose Nyquist column keeps only its real part in the C2R transform).  ``Di`` is the
    imaginary part the y-Nyquist frequency (f = -1/2, even FFT length) leaves behind.
    Gives the adjoint and the derivative w.r.t. the shift that the reference gets from
    autograd; the forward is checked against the FFT implementation in the tests."""

    def __init__(self, shape, shift):
        h, w = shape
        Fy, Fx = fft_shape(shape, shape, padding=10, axes=(-2, -1))
        self.fft_shape = (Fy, Fx)
        sy, sx = float(shift[0]), float(shift[1])
        dy = np.arange(-(h - 1), h)
        k = np.fft.fftfreq(Fy) * Fy  # integer frequencies, -Fy/2 at Nyquist
        ang = 2 * np.pi * k[None, :] * (dy[:, None] - sy) / Fy
        wk = 2 * np.pi * k / Fy
        dx = np.arange(-(w - 1), w)
        l = np.arange(Fx // 2 + 1)
        c = np.where((l == 0) | (l == Fx // 2), 1.0, 2.0)
        bng = 2 * np.pi * l[None, :] * (dx[:, None] - sx) / Fx
        wl = 2 * np.pi * l / Fx

        def toep(v, n):
            return v[np.arange(n)[:, None] - np.arange(n)[None, :] + (n - 1)]

        self.Dr = toep(np.cos(ang).sum(1) / Fy, h)
        self.Di = toep(np.sin(ang).sum(1) / Fy, h)
        self.dDr = toep((wk * np.sin(ang)).sum(1) / Fy, h)    # d/d sy
        self.dDi = toep(-(wk * np.cos(ang)).sum(1) / Fy, h)
        self.Tx = toep((c * np.cos(bng)).sum(1) / Fx, w)
        self.Hx = toep((c * np.sin(bng)).sum(1) / Fx, w)
        self.dTx = toep((c * wl * np.sin(bng)).sum(1) / Fx, w)  # d/d sx
        self.dHx = toep(-(c * wl * np.cos(bng)).sum(1) / Fx, w)

    def forward(self, x):
        return self.Dr @ x @ self.Tx.T - self.Di @ x @ self.Hx.T

    def adjoint(self, g):
        """Gradient w.r.t. the unshifted image given the gradient ``g`` w.r.t. the
        shifted one."""
        return self.Dr.T @ g @ self.Tx - self.Di.T @ g @ self.Hx

    def derivative_images(self, x):
        """(d/d sy, d/d sx) of ``forward(x)``."""
        return (self.dDr @ x @ self.Tx.T - self.dDi @ x @ self.Hx.T,
                self.Dr @ x @ self.dTx.T - self.Di @ x @ self.dHx.T)

    def shift_gradient(self, x, g):
        """(d/d sy, d/d sx) of ``sum(g * forward(x))``."""
        d_y = self.dDr @ x @ self.Tx.T - self.dDi @ x @ self.Hx.T
        d_x = self.Dr @ x @ self.dTx.T - self.Di @ x @ self.dHx.T
        return np.array([(g * d_y).sum(), (g * d_x).sum()])
