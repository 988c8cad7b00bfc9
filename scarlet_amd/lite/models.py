"""``scarlet.lite`` models (reference scarlet/lite/models.py): one observation on one
pixel grid, factorized components that carry their own optimizer parameters.

Same class names, constructor signatures and ``LiteBlend.fit`` semantics as the
reference; the loop itself -- model, PSF convolution, likelihood gradient, FISTA or
proximal-AMSGrad update, monotonicity with centre fitting, background threshold,
normalisation -- runs on the GPU through ``libscarlet_amd.so``.  The host steps in only
every ``resize`` iterations for ``LiteComponent.resize`` (lite/models.py:72-127).
"""

from functools import partial

import numpy as np

from .. import _lib, fft, initialization
from ..batch import BlendBatch, ComponentSpec
from ..bbox import Box, overlapped_slices
from ..constraint import MonotonicityConstraint
from ..parameter import relative_step
from .parameters import AdaproxParameter, FistaParameter
from .utils import insert_image


def _device_convolve(cube, kernel):
    """Zero-boundary 'same' convolution of a (C, h, w) cube with a (Ck, p, p) stamp on
    the GPU (the cube is presented to a one-blend batch as unit-spectrum components)."""
    cube = np.ascontiguousarray(cube, dtype=np.float32)
    C, H, W = cube.shape
    eye = np.eye(C, dtype=np.float32)
    comps = [ComponentSpec(eye[c], cube[c], (0, 0), prox_flags=0) for c in range(C)]
    ones = np.ones((1, C, H, W), dtype=np.float32)
    batch = BlendBatch(ones * 0, ones, [comps], kernel=np.ascontiguousarray(kernel, np.float32),
                       max_iter=1)
    try:
        return batch.forward(model=False)[1][0]
    finally:
        batch.close()


def _filter_bounds(shape):
    """Per tap of an odd (p, q) stamp the block offsets of the real-space convolution
    ``result[ys:, xs:] += v * image[ye:, xe:]`` (interpolation.py:7-65)."""
    p, q = shape
    if p % 2 == 0 or q % 2 == 0:
        raise ValueError("ambiguous centre: the stamp must have odd height and width")
    cy, cx = np.meshgrid(np.arange(p) - p // 2, np.arange(q) - q // 2, indexing="ij")
    cy, cx = cy.reshape(-1), cx.reshape(-1)
    zero = np.zeros_like(cy)
    return tuple(_lib.i32(a) for a in (np.maximum(zero, cy), -np.minimum(zero, cy),
                                        np.maximum(zero, cx), -np.minimum(zero, cx)))


def _real_convolve(cube, kernel):
    """Real-space convolution band by band with the tap loop of the reference's native
    ``apply_filter`` (renderer.py:97-117, operators_pybind11.cc:39-56) on the GPU
    (``smi_apply_filter_*``, bit-identical to the C++): exact zeros stay zeros."""
    import ctypes

    lib = _lib.load()
    dtype = np.float64 if cube.dtype == np.float64 else np.float32
    ct, fn = ((ctypes.c_double, lib.smi_apply_filter_f64) if dtype == np.float64
              else (ctypes.c_float, lib.smi_apply_filter_f32))
    cube = np.ascontiguousarray(cube, dtype=dtype)
    ys, ye, xs, xe = _filter_bounds(kernel.shape[1:])
    out = np.empty_like(cube)
    for c in range(cube.shape[0]):
        vals = np.ascontiguousarray(kernel[c if kernel.shape[0] > 1 else 0].reshape(-1), dtype=dtype)
        _lib.check(fn(_lib.ptr(cube[c], ct), cube.shape[1], cube.shape[2], _lib.ptr(vals, ct),
                      vals.size, _lib.ptr(ys, ctypes.c_int32), _lib.ptr(ye, ctypes.c_int32),
                      _lib.ptr(xs, ctypes.c_int32), _lib.ptr(xe, ctypes.c_int32),
                      _lib.ptr(out[c], ct)))
    return out


class LiteComponent:
    """Base component: centre, box, spectrum, morphology, background threshold
    (lite/models.py:19-133)."""

    def __init__(self, center, bbox, sed=None, morph=None, initialized=False, bg_thresh=0.25,
                 bg_rms=0):
        self._center = center
        self._bbox = bbox
        self._sed = sed
        self._morph = morph
        self.initialized = initialized
        self.bg_thresh = bg_thresh
        self.bg_rms = bg_rms

    center = property(lambda self: self._center)
    bbox = property(lambda self: self._bbox)
    sed = property(lambda self: self._sed)
    morph = property(lambda self: self._morph)

    def resize(self):
        """Shrink the (square) box when its outer rings are empty, grow it when the
        mean edge flux of the model exceeds ``bg_thresh * bg_rms`` in some band.
        Returns True if the box changed.  (The onion peeling compares row/column
        ``dist`` with ``-dist``, so its first step looks at row/column 0 twice --
        kept as in the reference, lite/models.py:84-91.)"""
        if self.bg_thresh is None:
            return False
        size = max(self.morph.shape)
        smaller = initialization.get_minimal_boxsize(size - 2 * self._empty_rings(self.morph))
        if smaller < size:
            inset = (size - smaller) // 2
            self._morph.shrink(inset)
            self._set_square(smaller, inset)
            return True
        model = self.get_model()
        edges = (model[:, 0], model[:, -1], model[0, :], model[-1, :])
        flux = np.array([np.sum(e) for e in edges])
        lit = np.array([np.sum(e > 0) for e in edges])
        with np.errstate(divide="ignore", invalid="ignore"):
            bright = np.any(flux / lit > self.bg_thresh * self.bg_rms[:, None, None])
        if not bright:
            return False
        larger = initialization.get_minimal_boxsize(size + 1)
        outset = (larger - size) // 2
        self._set_square(larger, -outset)
        self._morph.grow(self.bbox.shape[1:], outset)
        return True

    @staticmethod
    def _empty_rings(morph):
        """Number of rings the reference's peeling loop removes.  Its step ``d`` tests
        rows / columns ``d`` and ``-d``, i.e. index 0 twice at d = 0 and the LAST one
        only at d = 1, so with L empty leading and T empty trailing rows or columns it
        stops at ``min(L, T + 1)``."""
        filled = np.argwhere(np.asarray(morph) != 0)
        if len(filled) == 0:
            return max(morph.shape) // 2
        leading = filled.min(axis=0).min()
        trailing = (np.array(morph.shape) - 1 - filled.max(axis=0)).min()
        return int(min(leading, trailing + 1))

    def _set_square(self, side, inset):
        """Make the spatial box a square of ``side`` pixels whose corner sits ``inset``
        pixels further in (negative: further out) and refresh the placement slices."""
        band0, y0, x0 = self.bbox.origin
        self.bbox.origin = (band0, y0 + inset, x0 + inset)
        self.bbox.shape = (self.bbox.shape[0], side, side)
        self.slices = overlapped_slices(self.model_bbox, self.bbox)

    def __repr__(self):
        return type(self).__name__

    __str__ = __repr__


class LiteFactorizedComponent(LiteComponent):
    """Spectrum x morphology with lite parameters (lite/models.py:140-263)."""

    def __init__(self, sed, morph, center, bbox, model_bbox, bg_rms, bg_thresh=0.25, floor=1e-20,
                 fit_center_radius=1):
        super().__init__(center, bbox, sed, morph, initialized=True, bg_thresh=bg_thresh,
                         bg_rms=bg_rms)
        self.monotonicity = MonotonicityConstraint(neighbor_weight="angle", min_gradient=0,
                                                   fit_center_radius=fit_center_radius)
        self.floor = floor
        self.model_bbox = model_bbox
        self._sed.grad, self._sed.prox = self.grad_sed, self.prox_sed
        self._morph.grad, self._morph.prox = self.grad_morph, self.prox_morph
        self.slices = overlapped_slices(model_bbox, bbox)

    sed = property(lambda self: self._sed.x)
    morph = property(lambda self: self._morph.x)

    def get_model(self, bbox=None):
        model = self.sed[:, None, None] * self.morph[None, :, :]
        if bbox is not None:
            model = insert_image(bbox, self.bbox, model, dtype=self.morph.dtype)
        return model

    def _boxed(self, input_grad):
        boxed = np.zeros(self.bbox.shape, dtype=self.morph.dtype)
        boxed[self.slices[1]] = input_grad[self.slices[0]]
        return boxed

    def grad_sed(self, input_grad, sed, morph):
        return np.einsum("...jk,jk", self._boxed(input_grad), morph)

    def grad_morph(self, input_grad, morph, sed):
        return np.einsum("i,i...", sed, self._boxed(input_grad))

    def prox_sed(self, sed, prox_step=0):
        sed[sed < self.floor] = self.floor
        return sed

    def prox_morph(self, morph, prox_step=0):
        """Monotonicity (sweep on the GPU), background threshold or positivity, centre
        floor, unit maximum (lite/models.py:217-238)."""
        morph = self.monotonicity(morph, 0)
        if self.bg_thresh is not None:
            level = self.bg_rms * self.bg_thresh
            model = self.sed[:, None, None] * morph[None, :, :]
            morph[np.all(model < level[:, None, None], axis=0)] = 0
        else:
            morph[morph < 0] = 0
        center = (morph.shape[0] // 2, morph.shape[1] // 2)
        morph[center] = max(morph[center], self.floor)
        morph[:] = morph / morph.max()
        return morph

    def update(self, it, input_grad):
        raise NotImplementedError(
            "scarlet_amd updates lite components on the GPU inside LiteBlend.fit")


class LiteSource:
    """Components of one astrophysical object (lite/models.py:266-331)."""

    def __init__(self, components, dtype):
        self.components = components
        self.dtype = dtype
        self.flux = None
        self.flux_box = None

    n_components = property(lambda self: len(self.components))
    is_null = property(lambda self: len(self.components) == 0)

    @property
    def center(self):
        return None if self.is_null else self.components[0].center

    @property
    def bbox(self):
        if self.is_null:
            return Box((0, 0, 0))
        bbox = self.components[0].bbox
        for c in self.components[1:]:
            bbox = bbox | c.bbox
        return bbox

    def get_model(self, bbox=None, use_flux=False):
        if self.is_null:
            return 0
        if use_flux:
            return self.flux if bbox is None else insert_image(bbox, self.flux_box, self.flux)
        if bbox is None:
            bbox = self.bbox
        model = np.zeros(bbox.shape, dtype=self.dtype)
        for c in self.components:
            dst, src = overlapped_slices(bbox, c.bbox)
            model[dst] += c.get_model()[src]
        return model

    def __repr__(self):
        return "LiteSource<{}>".format(len(self.components))

    def __str__(self):
        return "LiteSource<{}>".format(",".join(str(c) for c in self.components))


class LiteObservation:
    """Images, variance, weights and PSFs on one pixel grid, with the difference kernel
    to the model PSF (lite/models.py:333-476)."""

    def __init__(self, images, variance, weights, psfs, model_psf=None, noise_rms=None, bbox=None,
                 padding=3, convolution_mode="fft"):
        self.images, self.variance, self.weights = images, variance, weights
        self.psfs = psfs if psfs.dtype == images.dtype else psfs.astype(images.dtype)
        assert convolution_mode in ["fft", "real"], "convolution_mode must be either 'fft' or 'real'"
        self.mode = convolution_mode
        if noise_rms is None:
            noise_rms = np.array(np.mean(np.sqrt(variance), axis=(1, 2)))
        self.noise_rms = noise_rms
        self.model_psf = model_psf
        self.padding = padding
        if model_psf is not None:
            self.diff_kernel = fft.match_psf(self.psfs, model_psf, padding=padding)
            # the gradient of a convolution is the convolution with the flipped kernel
            self.grad_kernel = fft.Fourier(self.diff_kernel.image[:, ::-1, ::-1])
        else:
            self.diff_kernel = self.grad_kernel = None
        self.bbox = Box(images.shape) if bbox is None else bbox

    def convolve(self, image, mode=None, grad=False):
        """Model -> observed seeing in every band on the GPU: "fft" through the batched
        FFT convolution, "real" through the tap loop of the native ``apply_filter``."""
        kernel = self.grad_kernel if grad else self.diff_kernel
        if kernel is None:
            return image
        mode = self.mode if mode is None else mode
        if mode not in ("fft", "real"):
            raise ValueError("mode must be either 'fft' or 'real', got {}".format(mode))
        if mode == "real":
            return _real_convolve(image, kernel.image).astype(image.dtype, copy=False)
        return _device_convolve(image, kernel.image).astype(image.dtype, copy=False)

    def render(self, model):
        return self.convolve(model)

    data = property(lambda self: self.images)
    shape = property(lambda self: self.images.shape)
    n_bands = property(lambda self: self.images.shape[0])
    dtype = property(lambda self: self.images.dtype)

    def __getitem__(self, i):
        images, variance, weights = self.images[i], self.variance[i], self.weights[i]
        psfs, noise_rms = self.psfs[i], self.noise_rms[i]
        if images.ndim == 2:
            images, variance, weights, psfs = images[None], variance[None], weights[None], psfs[None]
            noise_rms = np.array([noise_rms])
        return LiteObservation(images, variance, weights, psfs, model_psf=self.model_psf,
                               noise_rms=noise_rms, bbox=self.bbox, padding=self.padding,
                               convolution_mode=self.mode)


class LiteBlend:
    """Sources + observation, fitted jointly (lite/models.py:479-624)."""

    def __init__(self, sources, observation):
        self.sources = sources
        self.components = [c for src in sources for c in src.components]
        self.observation = observation
        self.it = 0
        self.loss = []

    bbox = property(lambda self: self.observation.bbox)

    def get_model(self, convolve=False, use_flux=False):
        model = np.zeros(self.bbox.shape, dtype=self.observation.images.dtype)
        if use_flux:
            for src in self.sources:
                dst, _ = overlapped_slices(self.bbox, src.flux_box)
                model[dst] += src.flux
            return model
        for c in self.components:
            model[c.slices[0]] += c.get_model()[c.slices[1]]
        return self.observation.convolve(model) if convolve else model

    @property
    def log_likelihood(self):
        return np.array(self.loss)

    def fit_spectra(self, clip=False):
        """Linear least-squares spectra for the current morphologies
        (lite/models.py:547-580)."""
        from .initialization import multifit_seds

        seds = multifit_seds(self.observation, [c.morph for c in self.components],
                             [c.bbox[1:] for c in self.components])
        for c, sed in zip(self.components, seds):
            c.sed[:] = sed
            c.sed[c.sed < 0] = 0
        if clip:
            keep = []
            for src in self.sources:
                src.components = [c for c in src.components if np.any(c.sed) > 0 and np.any(c.morph) > 0]
                keep += src.components
            self.components = keep
        else:
            for c in self.components:
                c.prox_sed(c.sed)
        return self

    # ------------------------------------------------------------------ device
    def _kind(self):
        kinds = {(type(c._sed), type(c._morph)) for c in self.components}
        if kinds == {(FistaParameter, FistaParameter)}:
            return "fista"
        if kinds == {(AdaproxParameter, AdaproxParameter)}:
            return "adaprox"
        raise NotImplementedError(
            "all components must use FistaParameter or all AdaproxParameter, got {}".format(kinds))

    def _spec(self, c, kind):
        if not isinstance(c, LiteFactorizedComponent):
            raise NotImplementedError("{} cannot be fitted on the device".format(type(c).__name__))
        if c.floor != 1e-20:
            raise NotImplementedError("only floor=1e-20 is supported on the device")
        mono = c.monotonicity
        if mono.fit_center_radius not in (0, 1) or mono.use_mask:
            raise NotImplementedError("fit_center_radius must be 0 or 1, use_mask False")
        flags = _lib.PROX_MONOTONIC | _lib.PROX_CENTER_ON | _lib.PROX_NORM_MAX
        flags |= _lib.PROX_FIT_CENTER if mono.fit_center_radius == 1 else 0
        kw = {}
        if c.bg_thresh is not None:
            kw["bg_level"] = np.asarray(c.bg_rms, dtype=np.float32) * np.float32(c.bg_thresh)
        else:
            flags |= _lib.PROX_POSITIVE
        if kind == "fista":
            if c._sed.step != c._morph.step:
                raise NotImplementedError("spectrum and morphology must share the FISTA step")
            kw["fista_step"] = float(c._sed.step)
        else:
            step = c._sed._step_spec
            if not (isinstance(step, partial) and step.func is relative_step
                    and step.keywords.get("axis") is None and not step.args):
                raise NotImplementedError("spectrum step must be partial(relative_step, ...)")
            if callable(c._morph._step_spec):
                raise NotImplementedError("morphology step must be a constant")
            kw.update(sed_rel_step=float(step.keywords.get("factor", 0.1)),
                      sed_min_step=np.asarray(step.keywords.get("minimum", 0), dtype=np.float32),
                      morph_step=float(c._morph._step_spec))
        return ComponentSpec(c.sed, c.morph, c.bbox.origin[1:], prox_flags=flags,
                             neighbor_weight=mono.neighbor_weight, min_gradient=mono.min_gradient,
                             center_floor=c.floor, **kw)

    def _optimizer(self, kind):
        """Batch-wide constants the components must agree on."""
        if kind == "fista":
            return dict(prox_max_iter=1, prox_e_rel=1e-6), None
        keys = {(p.scheme, p.b1[0], p.b2, p.eps, p.max_prox_iter, p.e_rel)
                for c in self.components for p in (c._sed, c._morph)}
        if len(keys) != 1:
            raise NotImplementedError("AdaproxParameters with different settings in one blend")
        scheme, b1, b2, eps, max_prox_iter, e_rel = keys.pop()
        if scheme != "amsgrad":
            raise NotImplementedError("only scheme='amsgrad' runs on the device")
        return dict(prox_max_iter=max_prox_iter, prox_e_rel=e_rel), dict(b1=b1, b2=b2, eps=eps)

    def _upload(self, kind, capacity):
        obs = self.observation
        if obs.diff_kernel is not None and any(s % 2 == 0 for s in obs.diff_kernel.image.shape[1:]):
            raise NotImplementedError("difference kernels need odd stamps (the flipped kernel "
                                      "of an even stamp is not the transposed convolution)")
        batch = BlendBatch(
            obs.images[None], obs.weights[None], [[self._spec(c, kind) for c in self.components]],
            kernel=None if obs.diff_kernel is None else obs.diff_kernel.image,
            max_iter=max(capacity, 1), scheme="fista" if kind == "fista" else "amsgrad",
            log_norm=False)
        comps = self.components
        if kind == "fista":
            batch.set_fista_state(z_sed=np.stack([c._sed.z for c in comps]),
                                  z_morph=[c._morph.z for c in comps],
                                  t=[(c._sed.t, c._morph.t) for c in comps])
        else:
            def finite(a):  # vhat starts at -inf (lite/parameters.py:267-269): any value
                return np.where(np.isfinite(a), a, 0)  # below v is equivalent

            batch.set_moments(
                m_sed=np.stack([c._sed.m for c in comps]), v_sed=np.stack([c._sed.v for c in comps]),
                vhat_sed=np.stack([finite(c._sed.vhat) for c in comps]),
                m_morph=[c._morph.m for c in comps], v_morph=[c._morph.v for c in comps],
                vhat_morph=[finite(c._morph.vhat) for c in comps])
        return batch

    def _download(self, batch, kind):
        seds, morphs = batch.parameters()
        if kind == "fista":
            st = batch.fista_state()
        else:
            st = batch.moments()
        for k, c in enumerate(self.components):
            c._sed.x = seds[k].astype(c._sed.x.dtype)
            c._morph.x = morphs[k].astype(c._morph.x.dtype)
            if kind == "fista":
                c._sed.z, c._morph.z = st["z_sed"][k].copy(), st["z_morph"][k].copy()
                c._sed.t, c._morph.t = float(st["t"][k][0]), float(st["t"][k][1])
            else:
                c._sed.m, c._sed.v, c._sed.vhat = (st[n][k].copy() for n in ("m_sed", "v_sed", "vhat_sed"))
                c._morph.m, c._morph.v, c._morph.vhat = (
                    st[n][k].copy() for n in ("m_morph", "v_morph", "vhat_morph"))

    def fit(self, max_iter, e_rel=1e-4, min_iter=1, resize=10, reweight=True):
        """Fit all parameters; returns ``(it, loss[-1])`` like the reference
        (lite/models.py:589-624): per iteration the likelihood gradient (loss appended,
        ``-1/2 sum w (d - m)^2``), the update of every component (spectrum first), every
        ``resize`` iterations the box check, then ``it > min_iter and |dL| < e_rel |L|``.
        The iteration counter persists in ``self.it`` across calls."""
        from .measure import weight_sources

        it = self.it
        converged = not self.components
        kind = None if converged else self._kind()
        while it < max_iter and not converged:
            settings, opt = self._optimizer(kind)
            if settings["prox_max_iter"] != 1 and settings["prox_e_rel"] != e_rel:
                raise NotImplementedError(
                    "more than one proximal sub-iteration needs prox_e_rel == e_rel")
            batch = self._upload(kind, max_iter - it)
            if opt:
                batch.set_optimizer(**opt)
            if self.loss:
                batch.set_previous_loss(-self.loss[-1])
            resized = False
            try:
                while it < max_iter and not converged and not resized:
                    # iterations up to and including the next one that ends with a box check
                    if resize is None:
                        last = max_iter - 1
                    else:
                        last = min(max(-(-it // resize), 1) * resize, max_iter - 1)
                    n = last - it + 1
                    before = len(batch.loss_history()[0])
                    batch.step(it, n, e_rel=e_rel, min_iter=min_iter,
                               prox_max_iter=settings["prox_max_iter"], check_convergence=True)
                    active, err = batch.status()
                    if err >= 0:
                        raise ArithmeticError("parameters of the blend are not finite")
                    n_done = len(batch.loss_history()[0]) - before
                    if active == 0:
                        # the stopping rule fired in iteration it + n_done - 1; the
                        # reference breaks before incrementing the counter
                        it += n_done - 1
                        converged = True
                    else:
                        it += n
                    ended = it if converged else it - 1
                    if resize is not None and ended > 0 and ended % resize == 0:
                        self._download(batch, kind)
                        resized = any([c.resize() for c in self.components
                                       if hasattr(c, "resize")])
                self.loss += [-float(v) for v in batch.loss_history()[0]]
                if not resized:
                    self._download(batch, kind)
            finally:
                batch.close()
        self.it = it
        if reweight:
            weight_sources(self)
        return it, self.loss[-1]
