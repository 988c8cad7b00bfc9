"""TEST INFRASTRUCTURE -- CPU restatement of the reference's multi-resolution renderer.

``ResolutionRenderer`` (scarlet/renderer.py:262-547), unrotated grids with the low-
resolution frame not wider than tall (``small_axis``; the branch every pair of
tests/test_multiresolution.py takes).  The set-up quantities that depend on WCS and PSF
interpolation -- the padded difference kernel, the sub-pixel positions of the low-
resolution rows / columns (``shifts``, ``other_shifts``), the FFT shape and the
pixel-scale ratio ``h`` -- are INPUTS here; the golden fixtures hold the reference's own
values for them.  What is restated is the arithmetic on the path:

* renderer.py:351-352, 414-476 (``sinc_shift`` along y): the operator
  ``resconv_op[c, a] = h^2 * (difference kernel Fourier-shifted along y to row a)``;
* renderer.py:478-545 (``transform``): centred zero padding of the model to the FFT shape,
  Fourier shift of every padded row to every low-resolution column ``b`` (by
  ``-other_shifts``), contraction ``out[c, a, b] = sum_yx resconv_op[c, a, y, x] *
  shifted[c, y, x, b]``.

A one-axis Fourier shift as the reference does it (``Fourier.fft``: ifftshift + rfft;
multiplication by ``exp(-2 pi i f s)`` with ``f = rfftfreq(F)``, interpolation.py:363-371;
``Fourier.from_fft``: irfft(F) + fftshift) is a real linear map on vectors of length F;
``shift_matrices`` returns it densely, which also gives the adjoint the gradient needs.
"""

import numpy as np


def shift_matrices(F, shifts):
    """(n, F, F) matrices ``M[k]`` with ``M[k] @ v`` = ``v`` Fourier-shifted by
    ``shifts[k]`` pixels (periodic, real transform of length ``F``)."""
    impulses = np.fft.ifftshift(np.eye(F), axes=0)
    spectrum = np.fft.rfft(impulses, axis=0)  # column j: spectrum of impulse j
    phase = np.exp(-2j * np.pi * np.fft.rfftfreq(F)[None, :] * np.asarray(shifts)[:, None])
    shifted = np.fft.irfft(phase[:, :, None] * spectrum[None], F, axis=1)
    return np.fft.fftshift(shifted, axes=1)


class LowResObservation:
    """A second observation of the model on a coarser grid, with its own data / weights.

    Parameters: ``kernel`` (C, Fy, Fx) padded difference kernel (``diff_kernel.image``),
    ``shifts_y`` (n_a,) = ``renderer.shifts[0]``, ``shifts_x`` (n_b,) =
    ``renderer.other_shifts[1]``, ``h`` pixel-scale ratio, ``channels`` = index of each of
    the C bands in the model cube, ``frame_hw`` spatial shape of the model frame.
    """

    def __init__(self, kernel, shifts_y, shifts_x, h, channels, frame_hw, data, weights):
        kernel = np.asarray(kernel, dtype=np.float64)
        self.C, self.Fy, self.Fx = kernel.shape
        self.channels = list(channels)
        self.frame_hw = tuple(frame_hw)
        self.data = np.asarray(data, dtype=np.float64)
        self.weights = np.asarray(weights, dtype=np.float64)
        # renderer.py:351-352: kernel shifted along y to every low-resolution row
        My = shift_matrices(self.Fy, shifts_y)  # (n_a, Fy, Fy)
        self.op = h**2 * np.einsum("ayz,czx->cayx", My, kernel)
        # renderer.py:498-505: rows shifted to every low-resolution column, by -other_shifts
        self.Mx = shift_matrices(self.Fx, -np.asarray(shifts_x))  # (n_b, Fx, Fx)
        # fft._pad (fft.py:82-113): centred embedding, start = (F - n + 1) // 2
        H, W = self.frame_hw
        self.y0, self.x0 = (self.Fy - H + 1) // 2, (self.Fx - W + 1) // 2

    def _pad(self, cube):
        out = np.zeros((self.C, self.Fy, self.Fx))
        H, W = self.frame_hw
        out[:, self.y0:self.y0 + H, self.x0:self.x0 + W] = cube
        return out

    def render(self, model):
        """(C_model, H, W) -> (C, n_a, n_b) (renderer.py:478-545, small_axis branch)."""
        padded = self._pad(np.asarray(model, dtype=np.float64)[self.channels])
        shifted = np.einsum("cyz,bxz->cyxb", padded, self.Mx)
        return np.einsum("cayx,cyxb->cab", self.op, shifted)

    def adjoint(self, upstream, n_model_channels):
        """Transpose of ``render``: (C, n_a, n_b) -> (C_model, H, W)."""
        back = np.einsum("cayx,cab->cyxb", self.op, upstream)
        padded = np.einsum("cyxb,bxz->cyz", back, self.Mx)
        H, W = self.frame_hw
        out = np.zeros((n_model_channels, H, W))
        out[self.channels] = padded[:, self.y0:self.y0 + H, self.x0:self.x0 + W]
        return out

    @property
    def log_norm(self):
        """``Observation.log_norm`` (observation.py:172-186)."""
        seen = self.weights > 0
        return seen.sum() / 2 * np.log(2 * np.pi) + np.sum(-0.5 * np.log(self.weights[seen]))

    def neg_log_likelihood(self, model):
        """``-Observation.get_log_likelihood`` (observation.py:147-170) and the upstream
        gradient ``w (m - d)`` with respect to the rendered image."""
        resid = self.render(model) - self.data
        return self.log_norm + 0.5 * np.sum(self.weights * resid**2), self.weights * resid

