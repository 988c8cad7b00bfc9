"""Oracle: the proximal-gradient loop of ``Blend.fit`` on plain NumPy arrays.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Scene description used here (no classes from the product):

* ``Component``: spectrum ``sed`` (C,), morphology ``morph`` (h, w), spatial
  ``origin`` (y0, x0) of the morphology box in model-frame pixels (may be
  negative or overhang the frame, as in the reference), prox/step settings and
  the AMSGrad moments of both parameters.
* ``Scene``: frame shape (C, H, W), ``data``/``weights`` cubes, the difference
  kernel stamp (Ck, P, P) with Ck in {1, C} or ``None`` for the NullRenderer.

dtype behaviour follows NumPy promotion exactly as the reference does: the
model cube is ``frame dtype`` (float32), morphologies are whatever the caller
passes (float64 in the reference after initialisation), moments are float64
(blend.py:155-160).
"""

import numpy as np

from . import fftconv, proxops


class Component:
    def __init__(
        self,
        sed,
        morph,
        origin,
        sed_min_step=0.0,
        morph_step=1e-2,
        monotonic="angle",
        min_gradient=0.0,
        symmetric=False,
        sed_zero=1e-20,
        source=None,
        shift=None,
        shift_step=1e-1,
        shift_rel_step=0.0,
        sparsity=None,
        tiny=1e-6,
        fixed=(False, False),
        state_dtype=np.float64,
    ):
        # Parameter(fixed=True) for (spectrum, image): it stays in X, autograd is not asked
        # for its gradient and adaprox sees zeros instead (blend.py:107-115); the step
        # (nothing) and the proximal operator are applied all the same
        self.fixed = tuple(fixed)
        # optional L0 / L1 member of the chain and the CenterOnConstraint floor
        self.sparsity, self.tiny = sparsity, tiny
        # ExtendedSourceMorphology(shifting=True): the sub-pixel offset of the centre
        # is a free 2-vector, step 1e-1, no constraint (morphology.py:673-676); the
        # model uses the Fourier-shifted image (morphology.py:124-130)
        self.shift = None if shift is None else np.array(shift, dtype=np.float64)
        # (a `relative_step` rule, parameter.py:126-129: max(shift_step, shift_rel_step * mean))
        self.shift_step, self.shift_rel_step = shift_step, shift_rel_step
        self.m_shift, self.v_shift, self.vhat_shift = np.zeros(2), np.zeros(2), np.zeros(2)
        self.sed = sed
        self.morph = morph
        self.origin = (int(origin[0]), int(origin[1]))
        # spectrum.py:54-56 -- relative_step(factor=1e-2, minimum=min_step)
        self.sed_min_step = sed_min_step
        # morphology.py:670 -- constant step
        self.morph_step = morph_step
        self.monotonic = monotonic
        self.min_gradient = min_gradient
        self.symmetric = symmetric
        self.sed_zero = sed_zero
        # components with the same (not None) `source` id form one
        # CombinedComponent, e.g. a MultiExtendedSource (source.py:615-717)
        self.source = source
        # blend.py:154-160: float64 zeros.  ``state_dtype=np.float32`` is NOT the
        # reference: it restates the device's arithmetic (float32 moments, float32
        # optimizer arithmetic) so that a test can tell how much of a difference between
        # the device and the reference-faithful oracle is the precision of the state and
        # how much anything else (tests/test_gpu_parity.py::test_hsc_fit_follows_*)
        self.state_dtype = np.dtype(state_dtype)
        self.m_sed = np.zeros(sed.shape, dtype=state_dtype)
        self.v_sed = np.zeros(sed.shape, dtype=state_dtype)
        self.vhat_sed = np.zeros(sed.shape, dtype=state_dtype)
        self.m_morph = np.zeros(morph.shape, dtype=state_dtype)
        self.v_morph = np.zeros(morph.shape, dtype=state_dtype)
        self.vhat_morph = np.zeros(morph.shape, dtype=state_dtype)

    def model_morph(self):
        """What enters the model: the image, Fourier-shifted if ``shift`` is free."""
        if self.shift is None:
            return self.morph
        return fftconv.fourier_shift(self.morph, self.shift)

    def sed_step(self, it=0):
        """``relative_step`` (parameter.py:126-129) with axis=None."""
        return np.maximum(self.sed_min_step, 1e-2 * self.sed.mean())

    def sed_prox(self, x, step):
        """``PositivityConstraint(zero=1e-20)`` (spectrum.py:54)."""
        return proxops.prox_positivity(x, step, self.sed_zero)

    def morph_prox(self, x, step):
        """ExtendedSourceMorphology chain (morphology.py:644-670)."""
        return proxops.morph_chain(
            x, step, self.monotonic, self.min_gradient, self.symmetric, self.sparsity, self.tiny,
            getattr(self, "chain_repeat", 1), getattr(self, "morph_zero", 0),
        )


class SameGridObservation:
    """A further observation on the model's pixel grid for ``Scene(extra_observations=)``:
    one more term of ``Blend._loss_func``'s sum over observations (blend.py:264-271) --
    e.g. a second exposure of bands the first observation already covers.  ``data`` /
    ``weights`` (C, H, W) over the model channels, ``kernel`` (Ck, P, P) or None."""

    def __init__(self, data, weights, kernel):
        self.data, self.weights, self.kernel = data, weights, kernel

    def render(self, model):
        return model if self.kernel is None else fftconv.convolve(model, self.kernel, axes=(1, 2))

    @property
    def log_norm(self):
        seen = self.weights != 0
        return seen.sum() / 2 * np.log(2 * np.pi) - 0.5 * np.sum(np.log(self.weights[seen]))

    def neg_log_likelihood(self, model):
        resid = self.render(model) - self.data
        return self.log_norm + 0.5 * np.sum(self.weights * resid**2), self.weights * resid

    def adjoint(self, upstream, n_model_channels):
        if self.kernel is None:
            return upstream
        return fftconv.convolve_adjoint(upstream, self.kernel, axes=(1, 2))


def integrated_gaussian(X, sigma):
    """Pixel-integrated 1-D Gaussian ``GaussianPSF._f`` (psf.py:128-142)."""
    from scipy.special import erfc

    sqrt2 = np.sqrt(2)
    return (
        np.sqrt(np.pi / 2)
        * sigma
        * (1 - erfc((0.5 - X) / (sqrt2 * sigma)) + 1 - erfc((2 * X + 1) / (2 * sqrt2 * sigma)))
    )


def integrated_gaussian_deriv(X, sigma):
    """d/dX of :func:`integrated_gaussian`: the difference of the Gaussian at the
    two pixel edges."""
    return np.exp(-((X + 0.5) ** 2) / (2 * sigma**2)) - np.exp(-((X - 0.5) ** 2) / (2 * sigma**2))


class PointComponent:
    """``PointSource`` (source.py:92-128): spectrum x model PSF evaluated at a
    free sub-pixel ``center``.  The box is the PSF box (psf.py:60-66) moved to
    the rounded initial centre (morphology.py:494-497) and never changes; the
    offset handed to the PSF is ``center - mean(box bounds)``
    (morphology.py:503-507).  ``morph`` is derived from ``center``."""

    source = None
    symmetric = False

    def __init__(self, sed, center, sigma, boxsize=None, sed_min_step=0.0, center_step=3e-2,
                 sed_zero=1e-20, center_rel_step=0.0, beta=0.0, image=None):
        self.sed = sed
        self.center = np.array(center, dtype=np.float64)
        # image: ImagePSF model PSF (psf.py:205-234): the stored image Fourier-shifted by the
        # offset (fft.shift, fft.py:399-428) -- `sigma` is not used then
        self.image = None if image is None else np.asarray(image, dtype=np.float64)
        if image is not None:
            # (a cube: an ImagePSF that differs between the bands -- the morphology is then a
            # cube too, one Fourier-shifted stamp per band, morphology.py:476-513)
            assert self.image.ndim in (2, 3) and self.image.shape[-2] == self.image.shape[-1]
            boxsize, sigma = self.image.shape[-1], 0.0
        self.sigma = float(sigma)
        # beta > 0: MoffatPSF(alpha=sigma, beta) instead of the Gaussian (psf.py:145-202)
        self.beta = float(beta)
        if boxsize is None:
            # psf.py:94-95 (Gaussian), psf.py:170-171 (Moffat)
            boxsize = int(np.ceil((5 if self.beta > 0 else 10) * self.sigma))
        if boxsize % 2 == 0:
            boxsize += 1  # psf.py:57-58
        self.size = boxsize
        pixel_center = np.round(self.center).astype(int)
        self.origin = (int(pixel_center[0]) - boxsize // 2, int(pixel_center[1]) - boxsize // 2)
        self.sed_min_step = sed_min_step
        self.center_step = center_step  # source.py:115
        self.center_rel_step = center_rel_step  # relative_step instead (parameter.py:126-129)
        self.sed_zero = sed_zero
        self.m_sed = np.zeros(sed.shape)
        self.v_sed = np.zeros(sed.shape)
        self.vhat_sed = np.zeros(sed.shape)
        self.m_center = np.zeros(2)
        self.v_center = np.zeros(2)
        self.vhat_center = np.zeros(2)

    def _axes(self):
        offset = self.center - (np.array(self.origin) + self.size / 2)
        grid = np.arange(self.size) - self.size // 2
        return grid - offset[0], grid - offset[1]

    @property
    def morph(self):
        """``GaussianPSF.get_model(offset=)`` for a band-independent sigma
        (psf.py:97-126): separable profile, normalised to unit sum."""
        if self.image is not None:
            offset = self.center - (np.array(self.origin) + self.size / 2)
            return fftconv.fourier_shift(self.image, offset)
        Y, X = self._axes()
        if self.beta > 0:  # MoffatPSF._f (psf.py:200-202), sampled at the pixel centres
            image = (1 + (X[None, :] ** 2 + Y[:, None] ** 2) / self.sigma**2) ** -self.beta
        else:
            image = integrated_gaussian(Y, self.sigma)[:, None] * integrated_gaussian(X, self.sigma)[None, :]
        return image / image.sum()

    def model_morph(self):
        return self.morph

    def center_gradient(self, g_morph):
        """Chain rule d(-logL)/d(center) = sum_yx g_morph * d(morph)/d(center)."""
        if self.image is not None:
            offset = self.center - (np.array(self.origin) + self.size / 2)
            op = fftconv.ShiftOperator(self.image.shape[-2:], offset)
            if self.image.ndim == 3:  # g_morph: (C, h, w), the band's spectrum already in it
                return sum(op.shift_gradient(img, g) for img, g in zip(self.image, g_morph))
            return op.shift_gradient(self.image, g_morph)
        Y, X = self._axes()
        if self.beta > 0:
            # A = q^-beta with q = 1 + r^2 / alpha^2; d A / d center = -d A / d (grid - offset)
            q = 1 + (X[None, :] ** 2 + Y[:, None] ** 2) / self.sigma**2
            A = q**-self.beta
            d = 2 * self.beta / self.sigma**2 * A / q
            dAy, dAx = d * Y[:, None], d * X[None, :]
            S = A.sum()
            return np.array([(g_morph * (dAy / S - A * dAy.sum() / S**2)).sum(),
                             (g_morph * (dAx / S - A * dAx.sum() / S**2)).sum()])
        fy, fx = integrated_gaussian(Y, self.sigma), integrated_gaussian(X, self.sigma)
        # d f(Y_j - offset)/d center = -f'(Y_j - offset)
        dfy, dfx = -integrated_gaussian_deriv(Y, self.sigma), -integrated_gaussian_deriv(X, self.sigma)
        S = fy.sum() * fx.sum()
        A = fy[:, None] * fx[None, :]
        d_y = dfy[:, None] * fx[None, :] / S - A * (dfy.sum() * fx.sum()) / S**2
        d_x = fy[:, None] * dfx[None, :] / S - A * (fy.sum() * dfx.sum()) / S**2
        return np.array([(g_morph * d_y).sum(), (g_morph * d_x).sum()])

    def sed_step(self, it=0):
        return np.maximum(self.sed_min_step, 1e-2 * self.sed.mean())

    def sed_prox(self, x, step):
        return proxops.prox_positivity(x, step, self.sed_zero)


def get_minimal_boxsize(size, min_size=21, increment=10):
    """initialization.py:173-177."""
    boxsize = min_size
    while boxsize < size:
        boxsize += increment
    return boxsize


def resize_component(c):
    """``ImageMorphology.update`` (morphology.py:132-207) on an oracle Component:
    shrink the (square) box when all four edges are empty, otherwise grow it when
    the next Adam step pulls flux over an edge.  Moments are sliced / zero-padded,
    the step is halved.  Returns True if the box changed (``UpdateException``)."""
    import numpy.ma as ma

    image = c.morph
    size = max(image.shape)
    # shrink_box (morphology.py:50-67)
    dist = 0
    while (
        np.all(image[dist, :] <= 0)
        and np.all(image[-dist - 1, :] <= 0)
        and np.all(image[:, dist] <= 0)
        and np.all(image[:, -dist - 1] <= 0)
    ):
        dist += 1
    newsize = get_minimal_boxsize(size - 2 * dist)
    if newsize < size:
        d = (size - newsize) // 2
        sl = (slice(d, d + newsize), slice(d, d + newsize))
        c.origin = (c.origin[0] + d, c.origin[1] + d)
        c.morph = image[sl]
        c.m_morph, c.v_morph, c.vhat_morph = c.m_morph[sl], c.v_morph[sl], c.vhat_morph[sl]
        c.morph_step = c.morph_step / 2
        return True
    gu = -c.m_morph / np.sqrt(np.sqrt(ma.masked_equal(c.v_morph, 0))) * c.morph_step
    pull = gu * (image > 0)
    edge_pull = np.array(
        (pull[:, 0].mean(), pull[:, -1].mean(), pull[0, :].mean(), pull[-1, :].mean())
    )
    if np.any(edge_pull > 0.1):
        newsize = get_minimal_boxsize(size + 1)
        pad = (newsize - size) // 2
        c.morph = np.pad(image, pad, mode="linear_ramp")
        c.m_morph = np.pad(c.m_morph, pad, mode="constant")
        c.v_morph = np.pad(c.v_morph, pad, mode="constant")
        c.vhat_morph = np.pad(c.vhat_morph, pad, mode="constant")
        c.morph_step = c.morph_step / 2
        c.origin = (c.origin[0] - pad, c.origin[1] - pad)
        return True
    return False


class Scene:
    def __init__(self, frame_shape, data, weights, kernel, components, dtype=np.float32,
                 psf_shift=None, extra_observations=(), psf_shift_step=1e-2, psf_shift_rel_step=0.0):
        # further observations of the same model (blend.py:265-271 sums their
        # log-likelihoods), e.g. oracle.resample.LowResObservation
        self.extra_observations = list(extra_observations)
        # ConvolutionRenderer(psf_shift=...) (renderer.py:175-177, 215-228): a free
        # sub-pixel shift of the difference kernel, step 1e-2, no constraint
        self.psf_shift = None if psf_shift is None else np.array(psf_shift, dtype=np.float64)
        self.m_psf, self.v_psf, self.vhat_psf = np.zeros(2), np.zeros(2), np.zeros(2)
        self.psf_shift_step, self.psf_shift_rel_step = psf_shift_step, psf_shift_rel_step
        # several observations with a free psf_shift each (blend.py:103-105 collects every
        # observation's parameters): `psf_groups` = [dict(bands=[...], shift=array(2), step=1e-2)],
        # one per observation -- its channels of the merged cube and its own shift
        self.psf_groups = None
        self.frame_shape = tuple(frame_shape)
        self.dtype = dtype
        self.data = data
        self.weights = weights
        self.kernel = kernel
        self.components = list(components)
        self.loss = []

    # -- forward ---------------------------------------------------------

    def box_slices(self, comp):
        """``overlapped_slices(frame.bbox, comp.bbox)`` on the two spatial axes
        (bbox.py:279-301; component.py:59-61)."""
        H, W = self.frame_shape[1:]
        h, w = comp.morph.shape[-2:]
        y0, x0 = comp.origin
        ylo, yhi = max(y0, 0), min(y0 + h, H)
        xlo, xhi = max(x0, 0), min(x0 + w, W)
        yhi, xhi = max(yhi, ylo), max(xhi, xlo)
        frame_sl = (slice(None), slice(ylo, yhi), slice(xlo, xhi))
        box_sl = (slice(None), slice(ylo - y0, yhi - y0), slice(xlo - x0, xhi - x0))
        return frame_sl, box_sl

    def _groups(self):
        groups = []
        for c in self.components:
            if groups and c.source is not None and groups[-1][0].source == c.source:
                groups[-1].append(c)
            else:
                groups.append([c])
        return groups

    def get_model(self):
        """``Blend.get_model`` (blend.py:200-244): scatter-add of every source's
        boxed model into a zero cube of frame dtype.  A single component is the
        outer product (component.py:160-164); a combined source first sums its
        children in float64 over their common box (component.py:254-278)."""
        full = np.zeros(self.frame_shape, dtype=self.dtype)
        H, W = self.frame_shape[1:]
        for group in self._groups():
            if len(group) == 1:
                c = group[0]
                mm = c.model_morph()
                boxed = c.sed[:, None, None] * (mm if mm.ndim == 3 else mm[None, :, :])
                fs, bs = self.box_slices(c)
                full[fs] += boxed[bs]
                continue
            y0 = min(c.origin[0] for c in group)
            x0 = min(c.origin[1] for c in group)
            y1 = max(c.origin[0] + c.morph.shape[0] for c in group)
            x1 = max(c.origin[1] + c.morph.shape[1] for c in group)
            summed = np.zeros((self.frame_shape[0], y1 - y0, x1 - x0))
            for c in group:
                oy, ox = c.origin[0] - y0, c.origin[1] - x0
                h, w = c.morph.shape
                summed[:, oy : oy + h, ox : ox + w] += c.sed[:, None, None] * c.model_morph()[None]
            ylo, yhi = max(y0, 0), min(y1, H)
            xlo, xhi = max(x0, 0), min(x1, W)
            if yhi > ylo and xhi > xlo:
                full[:, ylo:yhi, xlo:xhi] += summed[:, ylo - y0 : yhi - y0, xlo - x0 : xhi - x0]
        return full

    def render(self, model):
        """``Observation.render`` for NullRenderer / ConvolutionRenderer with
        identical model and data footprints (renderer.py:86-94, 247-259)."""
        if self.kernel is None:
            return model
        return fftconv.convolve(model, self.shifted_kernel(), axes=(1, 2))

    def shifted_kernel(self):
        """The difference kernel, Fourier-shifted by ``psf_shift`` and cropped back to
        its stamp (renderer.py:220-228 -> fft.py:399-428)."""
        if self.psf_groups:
            out = np.array(self.kernel, copy=True)
            for g in self.psf_groups:
                out[g["bands"]] = fftconv.fourier_shift(self.kernel[g["bands"]], g["shift"]).astype(out.dtype)
            return out
        if self.psf_shift is None:
            return self.kernel
        return fftconv.fourier_shift(self.kernel, self.psf_shift).astype(self.kernel.dtype)

    def render_adjoint(self, grad):
        if self.kernel is None:
            return grad
        return fftconv.convolve_adjoint(grad, self.shifted_kernel(), axes=(1, 2))

    def psf_shift_gradient(self, model, rendered):
        """d(-logL)/d(psf_shift) = sum w (m - d) * (model (*) d kernel / d shift)."""
        r = self.weights * (rendered - self.data)

        def gradient(bands, shift):
            op = fftconv.ShiftOperator(self.kernel.shape[1:], shift)
            dk = [np.stack([d[a] for d in (op.derivative_images(k.astype(np.float64))
                                           for k in self.kernel[bands])]) for a in range(2)]
            return np.array([np.sum(r[bands] * fftconv.convolve(
                model[bands], d.astype(self.kernel.dtype), axes=(1, 2))) for d in dk])

        if self.psf_groups:
            return [gradient(g["bands"], g["shift"]) for g in self.psf_groups]
        return gradient(slice(None), self.psf_shift)

    @property
    def log_norm(self):
        """``Observation.log_norm`` (observation.py:172-186)."""
        import numpy.ma as ma

        # observation.py:116-124: masked array, zeros of the weights masked
        noise_rms = 1 / np.sqrt(ma.masked_equal(self.weights, 0))
        n = np.prod(self.data.shape) - noise_rms.mask.sum()
        with np.errstate(divide="ignore"):
            return n / 2 * np.log(2 * np.pi) + np.log(noise_rms).sum()

    def log_likelihood(self, rendered):
        """``Observation.get_log_likelihood`` (observation.py:147-170)."""
        return -self.log_norm - np.sum(self.weights * (rendered - self.data) ** 2) / 2

    # -- gradient --------------------------------------------------------

    def model_gradient(self, rendered):
        """d(-logL)/d(model cube): ``w (m - d)`` pulled back through the
        renderer (lite/models.py:537-545)."""
        return self.render_adjoint(self.weights * (rendered - self.data))

    def parameter_gradients(self, G):
        """Per component d(-logL)/d(sed), d(-logL)/d(morph): slice ``G`` into
        the box (zero where the box overhangs the frame; blend.py:30-46), then
        ``sum_yx G*morph`` and ``sum_c sed*G`` (lite/models.py:206-216)."""
        out = []
        for c in self.components:
            h, w = c.morph.shape[-2:]
            boxed = np.zeros((self.frame_shape[0], h, w), dtype=np.float64)
            fs, bs = self.box_slices(c)
            boxed[bs] = G[fs]
            mm = c.model_morph()
            if mm.ndim == 3:  # a point source on a band-dependent model PSF
                g_sed = np.einsum("cyx,cyx->c", boxed, mm)
                g_morph = c.sed[:, None, None] * boxed
            else:
                g_sed = np.einsum("cyx,yx->c", boxed, mm)
                g_morph = np.einsum("c,cyx->yx", c.sed, boxed)
            fixed_sed, fixed_morph = getattr(c, "fixed", (False, False))
            if fixed_sed:
                g_sed = np.zeros_like(g_sed)
            if isinstance(c, PointComponent):
                # second entry is d/d(center) for a point source
                g_morph = c.center_gradient(g_morph)
                if fixed_morph:
                    g_morph = np.zeros_like(g_morph)
            elif c.shift is not None:
                # pull the gradient back through the Fourier shift; third entry d/d(shift)
                op = fftconv.ShiftOperator(c.morph.shape, c.shift)
                pulled = np.zeros_like(g_morph) if fixed_morph else op.adjoint(g_morph)
                out.append((g_sed, pulled, op.shift_gradient(c.morph, g_morph)))
                continue
            if fixed_morph and not isinstance(c, PointComponent):
                g_morph = np.zeros_like(g_morph)
            out.append((g_sed, g_morph))
        return out

    def loss_and_gradients(self):
        """One evaluation of ``Blend._loss_func`` (blend.py:259-274) and of its
        gradient; appends the loss like the reference does."""
        model = self.get_model()
        rendered = self.render(model)
        loss = -self.log_likelihood(rendered)
        G = self.model_gradient(rendered)
        for obs in self.extra_observations:
            term, upstream = obs.neg_log_likelihood(model)
            loss = loss + term
            G = G + obs.adjoint(upstream, self.frame_shape[0])
        self.loss.append(loss)
        if self.psf_shift is not None or self.psf_groups:
            self.g_psf_shift = self.psf_shift_gradient(model, rendered)
        return loss, self.parameter_gradients(G)

    # -- optimizer -------------------------------------------------------

    def step(self, it, e_rel, prox_max_iter=10, b1=0.9, b2=0.999, eps=1e-8):
        """One iteration of ``proxmin.adaprox`` as configured at
        blend.py:165-180 (amsgrad, prox_max_iter=10)."""
        _, grads = self.loss_and_gradients()
        # all steps are evaluated on the pre-update parameters (blend.py:135-138)
        # A free 2-vector (centre, shift, psf_shift) is a Parameter like any other
        # (blend.py:120-145): `vec_rules[name]`, if set on the component / scene, holds
        # {"prior": f(x) -> array added to the gradient (blend.py:120-131), "prox": f(x, step),
        # "step": f(x, it=...)} for it.
        def rules(owner, name):
            return (getattr(owner, "vec_rules", None) or {}).get(name, {})

        def vec_alpha(owner, name, x, default):
            fn = rules(owner, name).get("step")
            return default if fn is None else fn(x, it=it)

        def vec_grad(owner, name, x, g):
            fn = rules(owner, name).get("prior")
            return g if fn is None else g + fn(x)

        alphas = [(c.sed_step(it),
                   vec_alpha(c, "center", c.center,
                             relative_step(c.center, c.center_rel_step, c.center_step))
                   if isinstance(c, PointComponent) else c.morph_step)
                  for c in self.components]
        shift_alphas = [None if getattr(c, "shift", None) is None else
                        vec_alpha(c, "shift", c.shift,
                                  relative_step(c.shift, c.shift_rel_step, c.shift_step))
                        for c in self.components]
        if self.psf_shift is not None:
            psf_alpha = vec_alpha(self, "psf_shift", self.psf_shift,
                                  relative_step(self.psf_shift, self.psf_shift_rel_step,
                                                self.psf_shift_step))
        if self.psf_groups:  # (steps on the pre-update shifts, like every other step)
            group_alphas = [relative_step(g["shift"], g.get("rel", 0.0), g.get("step", 1e-2))
                            for g in self.psf_groups]
        if self.psf_shift is not None:
            # the renderer's parameter comes after the sources' in X (blend.py:103-105)
            g_psf = vec_grad(self, "psf_shift", self.psf_shift, self.g_psf_shift)
        for c, (g_sed, g_morph, *g_shift), (a_sed, a_morph), a_shift in zip(
                self.components, grads, alphas, shift_alphas):
            adaprox_update(
                it, c.sed, g_sed, c.m_sed, c.v_sed, c.vhat_sed, a_sed, c.sed_prox,
                e_rel, prox_max_iter, b1, b2, eps,
            )
            if isinstance(c, PointComponent):
                # the centre has no constraint by default (source.py:115): plain AMSGrad step
                adaprox_update(
                    it, c.center, vec_grad(c, "center", c.center, g_morph), c.m_center,
                    c.v_center, c.vhat_center, a_morph, rules(c, "center").get("prox"), e_rel,
                    prox_max_iter, b1, b2, eps,
                )
                continue
            adaprox_update(
                it, c.morph, g_morph, c.m_morph, c.v_morph, c.vhat_morph, a_morph,
                c.morph_prox, e_rel, prox_max_iter, b1, b2, eps,
            )
            if g_shift:
                adaprox_update(
                    it, c.shift, vec_grad(c, "shift", c.shift, g_shift[0]), c.m_shift, c.v_shift,
                    c.vhat_shift, a_shift, rules(c, "shift").get("prox"), e_rel, prox_max_iter,
                    b1, b2, eps,
                )
        if self.psf_shift is not None:
            adaprox_update(it, self.psf_shift, g_psf, self.m_psf, self.v_psf, self.vhat_psf, psf_alpha,
                           rules(self, "psf_shift").get("prox"), e_rel, prox_max_iter, b1, b2, eps)
        if self.psf_groups:
            for g, alpha, grad in zip(self.psf_groups, group_alphas, self.g_psf_shift):
                for name in ("m", "v", "vhat"):
                    g.setdefault(name, np.zeros(2))
                adaprox_update(it, g["shift"], grad, g["m"], g["v"], g["vhat"], alpha, None, e_rel,
                               prox_max_iter, b1, b2, eps)

    def check_parameters(self):
        """``Model.check_parameters`` (model.py:153-165)."""
        for k, c in enumerate(self.components):
            free = c.center if isinstance(c, PointComponent) else c.morph
            if not (np.isfinite(c.sed).all() and np.isfinite(free).all()):
                raise ArithmeticError("component {} is not finite".format(k))

    def fit(self, max_iter=200, e_rel=1e-3, min_iter=1, prox_max_iter=10, resizing=False,
            b1=0.9, b2=0.999, eps=1e-8):
        """``Blend.fit`` (blend.py:85-198).

        Iteration order follows the reference authors' own loop
        (lite/models.py:589-624): gradient (loss appended) -> parameter updates
        -> [every 10 iterations the resize hook, blend.py:284-292] -> convergence
        test ``it > min_iter and |dL| < e_rel |L|``.  A resize restarts the
        optimizer counter (``UpdateException``, blend.py:196-198): the next
        iteration is again a tenth of a step with ``vhat = v``.  The position of
        proxmin's callback inside ``adaprox`` is third-party code (parity
        unpinned, see oracle/__init__.py).

        Returns ``(len(loss), logL[-1])`` like blend.py:194.
        """
        it = 0
        while it < max_iter:
            local = 0
            restart = False
            while it + local < max_iter:
                self.step(local, e_rel, prox_max_iter, b1, b2, eps)
                self.check_parameters()
                if resizing and local > 0 and local % 10 == 0:
                    # Blend._callback calls src.update() per source (blend.py:284-292);
                    # a combined source stops at its first child that resizes
                    # (component.py:280-290 re-raises out of the loop)
                    for group in self._groups():
                        for c in group:
                            if not isinstance(c, PointComponent) and resize_component(c):
                                restart = True
                                break
                    if restart:
                        break
                if local > min_iter and abs(self.loss[-1] - self.loss[-2]) < e_rel * abs(
                    self.loss[-1]
                ):
                    return len(self.loss), -self.loss[-1]
                local += 1
            if not restart:
                break
            it = len(self.loss)
        return len(self.loss), -self.loss[-1]


def relative_step(X, factor, minimum):
    """``relative_step`` (parameter.py:126-129) with ``axis=None``: ``max(minimum, factor *
    X.mean())``; ``factor = 0`` stands for a constant step ``minimum``."""
    return max(minimum, factor * float(np.mean(X))) if factor else minimum


def l2sq(x):
    """``proxmin.utils.l2sq``: squared Euclidean norm."""
    return (x**2).sum()


def amsgrad_phi_psi(it, g, m, v, vhat, b1, b2, eps):
    """AMSGrad moments (Reddi, Kale & Kumar 2018) as used by
    ``proxmin.algorithms._amsgrad_phi_psi``: no bias correction, ``vhat = v``
    on the first iteration, floor ``eps`` under the square root.
    In-place on ``m, v, vhat`` (warm start, blend.py:153-163)."""
    m[:] = (1 - b1) * g + b1 * m
    v[:] = (1 - b2) * (g**2) + b2 * v
    if it == 0:
        vhat[:] = v
    else:
        vhat[:] = np.maximum(vhat, v)
    psi = np.sqrt(np.maximum(vhat, eps)) if eps > 0 else np.sqrt(vhat)
    return m, psi


def adaprox_update(it, x, g, m, v, vhat, alpha, prox, e_rel, prox_max_iter=10,
                   b1=0.9, b2=0.999, eps=1e-8):
    """Inner update of one parameter (lite/parameters.py:274-305): gradient
    step ``x -= alpha phi/psi`` (a tenth of it on the first iteration), then
    up to ``prox_max_iter`` proximal sub-iterations in the metric ``psi`` with
    step ``gamma = alpha / max(psi)``, stopped when the relative squared
    change drops to ``e_rel**2``.  ``x`` is updated in place."""
    if m.dtype == np.float32:
        # float32-state mode (device arithmetic, see Component): keep NumPy from
        # promoting the update to float64 through a float64 gradient / step
        g = np.asarray(g, dtype=np.float32)
        alpha = np.float32(alpha) if np.ndim(alpha) == 0 else np.asarray(alpha, np.float32)
        b1, b2, eps = np.float32(b1), np.float32(b2), np.float32(eps)
    phi, psi = amsgrad_phi_psi(it, g, m, v, vhat, b1, b2, eps)
    if it > 0:
        x -= alpha * phi / psi
    else:
        x -= alpha * phi / psi / 10
    if prox is not None:
        z = x.copy()
        gamma = alpha / np.max(psi)
        for _tau in range(1, prox_max_iter + 1):
            z_new = prox(z - gamma / alpha * psi * (z - x), gamma)
            converged = l2sq(z_new - z) <= e_rel**2 * l2sq(z)
            z = z_new
            if converged:
                break
        x[...] = z
    return x
