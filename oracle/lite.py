"""Oracle: the ``scarlet.lite`` fitting loop (reference scarlet/lite/models.py,
scarlet/lite/parameters.py).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  NumPy restatement of
``LiteBlend.fit`` with ``LiteFactorizedComponent`` and the two parameter classes.

Pinning: ``FistaParameter`` is entirely in-repo code of the reference (no proxmin, no
autograd), so the golden ``tests/golden/lite_fista.npz`` holds a trajectory the
reference itself ran in the build container (``oracle/refshim/make_golden.py``) and
this restatement is checked against it iteration by iteration.  ``AdaproxParameter``
calls ``proxmin.algorithms._amsgrad_phi_psi`` (third party, absent): its golden was
run with the reference's own ``AdaproxParameter.update`` / ``LiteBlend.fit`` around the
shim's five-line AMSGrad moments, which narrows "parity unpinned" to those lines.
"""

import numpy as np

from . import fftconv, proxops
from .pgm import amsgrad_phi_psi, get_minimal_boxsize, l2sq


def get_center(image, center, radius=1):
    """Brightest pixel within ``radius`` of ``center`` (operator.py:99-129)."""
    cy, cx = int(center[0]), int(center[1])
    y0, x0 = max(cy - radius, 0), max(cx - radius, 0)
    subset = image[y0 : cy + radius + 1, x0 : cx + radius + 1]
    dy, dx = np.unravel_index(np.argmax(subset), subset.shape)
    return dy + y0, dx + x0


def monotonicity(morph, fit_center_radius=1, neighbor_weight="angle", min_gradient=0):
    """``MonotonicityConstraint.__call__`` (constraint.py:203-234) as configured by
    ``LiteFactorizedComponent`` (lite/models.py:176-180); in place."""
    center = (morph.shape[0] // 2, morph.shape[1] // 2)
    if fit_center_radius > 0:
        center = get_center(morph, center, fit_center_radius)
    weights, didx, offsets = proxops.monotonic_operator(morph.shape, neighbor_weight, center)
    proxops.sweep(morph, weights, offsets, didx, min_gradient)
    return morph


class LiteComponent:
    """``LiteFactorizedComponent`` (lite/models.py:140-263) with its two parameters.

    ``kind`` = "fista": ``FistaParameter`` (lite/parameters.py:92-165), step
    ``fista_step`` (lite/initialization.py:308-312); "adaprox": ``AdaproxParameter``
    (lite/parameters.py:185-317) with the steps of ``init_adaprox_component``
    (lite/initialization.py:250-284): spectrum ``relative_step(factor=1e-2,
    minimum=noise_rms/factor)``, morphology 1e-2, ``max_prox_iter`` = 1."""

    def __init__(self, sed, morph, origin, bg_rms, bg_thresh=0.25, floor=1e-20, kind="fista",
                 fista_step=None, sed_min_step=0.0, morph_step=1e-2, max_prox_iter=1,
                 prox_e_rel=1e-6, fit_center_radius=1):
        self.sed, self.morph = sed, morph
        self.origin = (int(origin[0]), int(origin[1]))
        self.bg_rms = np.asarray(bg_rms)
        self.bg_thresh = bg_thresh
        self.floor = floor
        self.kind = kind
        self.fit_center_radius = fit_center_radius
        if kind == "fista":
            self.fista_step = fista_step
            self.z_sed, self.z_morph = sed, morph  # z0 = x (lite/parameters.py:126-127)
            self.t_sed = self.t_morph = 1
        else:
            self.sed_min_step, self.morph_step = sed_min_step, morph_step
            self.max_prox_iter, self.prox_e_rel = max_prox_iter, prox_e_rel
            self.m_sed, self.v_sed = np.zeros(sed.shape, sed.dtype), np.zeros(sed.shape, sed.dtype)
            self.vhat_sed = np.full(sed.shape, -np.inf, dtype=sed.dtype)
            self.m_morph = np.zeros(morph.shape, morph.dtype)
            self.v_morph = np.zeros(morph.shape, morph.dtype)
            self.vhat_morph = np.full(morph.shape, -np.inf, dtype=morph.dtype)

    def get_model(self):
        return self.sed[:, None, None] * self.morph[None, :, :]

    # -- proximal operators (lite/models.py:209-238) ---------------------------
    def prox_sed(self, sed, step=0):
        sed[sed < self.floor] = self.floor
        return sed

    def prox_morph(self, morph, step=0):
        morph = monotonicity(morph, self.fit_center_radius)
        if self.bg_thresh is not None:
            level = self.bg_rms * self.bg_thresh
            model = self.sed[:, None, None] * morph[None, :, :]
            morph[np.all(model < level[:, None, None], axis=0)] = 0
        else:
            morph[morph < 0] = 0
        center = (morph.shape[0] // 2, morph.shape[1] // 2)
        morph[center] = np.max([morph[center], self.floor])
        morph[:] = morph / morph.max()
        return morph

    # -- parameter updates ------------------------------------------------------
    def _fista(self, x, z, t, g, other, prox):
        """lite/parameters.py:134-150"""
        step = self.fista_step / np.sum(other * other)
        y = z - step * g
        x_new = prox(y, step)
        t_new = 0.5 * (1 + np.sqrt(1 + 4 * t**2))
        omega = 1 + (t - 1) / t_new
        return x_new, x + omega * (x_new - x), t_new

    def _adaprox(self, it, x, g, m, v, vhat, step, prox):
        """lite/parameters.py:274-305 with scheme 'amsgrad'"""
        phi, psi = amsgrad_phi_psi(it, g, m, v, vhat, 0.9, 0.999, 1e-8)
        if it > 0:
            x -= step * phi / psi
        else:
            x -= step * phi / psi / 10
        z = x.copy()
        gamma = step / np.max(psi)
        for _ in range(1, self.max_prox_iter + 1):
            z_new = prox(z - gamma / step * psi * (z - x), gamma)
            converged = l2sq(z_new - z) <= self.prox_e_rel**2 * l2sq(z)
            z = z_new
            if converged:
                break
        return z

    def update(self, it, g_sed, g_morph):
        """``LiteFactorizedComponent.update`` (lite/models.py:240-247): the spectrum
        first (with the current morphology), then the morphology with the *old*
        spectrum in its gradient and step; its threshold sees the new spectrum."""
        sed_old = self.sed.copy()
        if self.kind == "fista":
            self.sed, self.z_sed, self.t_sed = self._fista(
                self.sed, self.z_sed, self.t_sed, g_sed, self.morph, self.prox_sed)
            self.morph, self.z_morph, self.t_morph = self._fista(
                self.morph, self.z_morph, self.t_morph, g_morph, sed_old, self.prox_morph)
        else:
            step = np.maximum(self.sed_min_step, 1e-2 * self.sed.mean())
            self.sed = self._adaprox(it, self.sed, g_sed, self.m_sed, self.v_sed, self.vhat_sed,
                                     step, self.prox_sed)
            self.morph = self._adaprox(it, self.morph, g_morph, self.m_morph, self.v_morph,
                                       self.vhat_morph, self.morph_step, self.prox_morph)

    # -- resizing (lite/models.py:72-127) ----------------------------------------
    def _state_arrays(self):
        names = ["morph"]
        names += ["z_morph"] if self.kind == "fista" else ["m_morph", "v_morph", "vhat_morph"]
        return names

    def resize(self):
        if self.bg_thresh is None:
            return False
        morph = self.morph
        size = max(morph.shape)
        dist = 0
        # note the reference's `-dist` (not -dist - 1): row/column -0 is row/column 0
        while (np.all(morph[dist, :] == 0) and np.all(morph[-dist, :] == 0)
               and np.all(morph[:, dist] == 0) and np.all(morph[:, -dist] == 0)):
            dist += 1
        new_size = get_minimal_boxsize(size - 2 * dist)
        if new_size < size:
            dist = (size - new_size) // 2
            self.origin = (self.origin[0] + dist, self.origin[1] + dist)
            for name in self._state_arrays():
                setattr(self, name, getattr(self, name)[dist:-dist, dist:-dist])
            return True
        model = self.get_model()
        edge_flux = np.array([np.sum(model[:, 0]), np.sum(model[:, -1]),
                              np.sum(model[0, :]), np.sum(model[-1, :])])
        edge_mask = np.array([np.sum(model[:, 0] > 0), np.sum(model[:, -1] > 0),
                              np.sum(model[0, :] > 0), np.sum(model[-1, :] > 0)])
        with np.errstate(divide="ignore", invalid="ignore"):
            grow = np.any(edge_flux / edge_mask > self.bg_thresh * self.bg_rms[:, None, None])
        if grow:
            new_size = get_minimal_boxsize(size + 1)
            dist = (new_size - size) // 2
            self.origin = (self.origin[0] - dist, self.origin[1] - dist)
            for name in self._state_arrays():
                old = getattr(self, name)
                new = np.zeros((new_size, new_size), dtype=old.dtype)
                new[dist:-dist, dist:-dist] = old
                setattr(self, name, new)
            return True
        return False


class LiteScene:
    """``LiteBlend`` + ``LiteObservation`` (lite/models.py:333-624)."""

    def __init__(self, images, weights, kernel, components):
        self.images, self.weights, self.kernel = images, weights, kernel
        self.components = list(components)
        self.it = 0
        self.loss = []

    def slices(self, c):
        H, W = self.images.shape[1:]
        h, w = c.morph.shape
        y0, x0 = c.origin
        ylo, yhi, xlo, xhi = max(y0, 0), min(y0 + h, H), max(x0, 0), min(x0 + w, W)
        yhi, xhi = max(yhi, ylo), max(xhi, xlo)
        return ((slice(None), slice(ylo, yhi), slice(xlo, xhi)),
                (slice(None), slice(ylo - y0, yhi - y0), slice(xlo - x0, xhi - x0)))

    def get_model(self):
        model = np.zeros(self.images.shape, dtype=self.images.dtype)
        for c in self.components:
            fs, bs = self.slices(c)
            model[fs] += c.get_model()[bs]
        return model

    def convolve(self, image, grad=False):
        """LiteObservation.convolve, 'fft' mode (lite/models.py:381-413); the gradient
        uses the flipped kernel (lite/models.py:363-367)."""
        if self.kernel is None:
            return image
        kernel = self.kernel[:, ::-1, ::-1] if grad else self.kernel
        return fftconv.convolve(image, kernel, axes=(1, 2))

    def grad_logL(self):
        """lite/models.py:537-545"""
        model = self.convolve(self.get_model())
        self.loss.append(0.5 * -np.sum(self.weights * (self.images - model) ** 2))
        return self.convolve(self.weights * (model - self.images), grad=True)

    def component_gradients(self, c, G):
        """grad_sed / grad_morph (lite/models.py:197-207)"""
        boxed = np.zeros((self.images.shape[0],) + c.morph.shape, dtype=c.morph.dtype)
        fs, bs = self.slices(c)
        boxed[bs] = G[fs]
        return np.einsum("...jk,jk", boxed, c.morph), np.einsum("i,i...", c.sed, boxed)

    def fit(self, max_iter, e_rel=1e-4, min_iter=1, resize=10):
        """``LiteBlend.fit`` (lite/models.py:589-624) without the final re-weighting."""
        it = self.it
        while it < max_iter:
            G = self.grad_logL()
            for c in self.components:
                g_sed, g_morph = self.component_gradients(c, G)
                c.update(it, g_sed, g_morph)
            if resize is not None and it > 0 and it % resize == 0:
                for c in self.components:
                    c.resize()
            if it > min_iter and np.abs(self.loss[-1] - self.loss[-2]) < e_rel * np.abs(self.loss[-1]):
                break
            it += 1
        self.it = it
        return it, self.loss[-1]
