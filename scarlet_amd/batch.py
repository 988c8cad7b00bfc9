"""``BlendBatch``: many independent blends fitted together on one GPU.

This is the batched form of ``Blend.fit`` (reference blend.py:85-198) that the
benchmark and the multi-GPU driver use; ``scarlet_amd.Blend.fit`` is a batch of
one.  All arithmetic happens in ``libscarlet_amd.so`` (HIP); this class only
packs NumPy arrays for the C ABI.
"""

import ctypes

import numpy as np

from . import _lib, operator


class ComponentSpec:
    """Plain description of one factorized component for the device loop."""

    def __init__(self, sed, morph, origin, sed_min_step=0.0, sed_rel_step=1e-2,
                 morph_step=1e-2, morph_rel_step=0.0, prox_flags=_lib.PROX_EXTENDED_SOURCE,
                 neighbor_weight="angle", min_gradient=0.0, l_thresh=0.0, shift=None,
                 shift_step=1e-1, center_floor=1e-6, bg_level=None, fista_step=0.0,
                 sym_strength=1.0, chain_repeat=1, pos_floor=0.0, shift_rel_step=0.0):
        self.sed = np.asarray(sed, dtype=np.float32)
        self.morph = np.ascontiguousarray(morph, dtype=np.float32)
        self.origin = (int(origin[0]), int(origin[1]))
        step = np.asarray(sed_min_step, dtype=np.float32)
        self.sed_min_step = step if step.shape == self.sed.shape else \
            np.broadcast_to(step, self.sed.shape)
        self.sed_rel_step = float(sed_rel_step)
        self.morph_step = float(morph_step)
        self.morph_rel_step = float(morph_rel_step)
        self.prox_flags = int(prox_flags)
        self.neighbor_weight = neighbor_weight
        self.min_gradient = float(min_gradient)
        self.l_thresh = float(l_thresh)
        # ExtendedSource(shifting=True): free sub-pixel Fourier shift of the image
        # (morphology.py:124-130, 673-676); the device keeps it in the `center` slot
        self.shift_step = float(shift_step)
        self.shift_rel_step = float(shift_rel_step)  # relative_step factor (parameter.py:126-129)
        # scarlet.lite: floor of the centre pixel, background threshold per band
        # (bg_rms * bg_thresh; sets PROX_BG_THRESH), FistaParameter.step
        self.center_floor = float(center_floor)
        self.bg_level = None
        if bg_level is not None:
            self.bg_level = np.broadcast_to(np.asarray(bg_level, dtype=np.float32), self.sed.shape)
            self.prox_flags |= _lib.PROX_BG_THRESH
        self.fista_step = float(fista_step)
        self.sym_strength = float(sym_strength)  # SymmetryConstraint(strength)
        self.chain_repeat = int(chain_repeat)  # ConstraintChain(repeat)
        self.pos_floor = float(pos_floor)  # PositivityConstraint(zero) of the morphology
        if shift is not None:
            self.center = np.array(shift, dtype=np.float64).reshape(2)
            self.prox_flags |= _lib.COMPONENT_SHIFTING


class PointSourceSpec(ComponentSpec):
    """A ``PointSource`` (source.py:92-128): spectrum x the model PSF -- a
    pixel-integrated Gaussian of width ``psf_sigma`` in every band, or with ``psf_beta`` > 0
    the Moffat profile ``(1 + r^2 / psf_sigma^2)^-psf_beta`` sampled at the pixel centres
    (psf.py:145-202) -- evaluated at the free sub-pixel ``center`` (frame pixels).  The box is
    the PSF box (psf.py:55-66, 93-95, 170-171) moved to the rounded initial centre
    (morphology.py:494-497) and stays fixed."""

    def __init__(self, sed, center, psf_sigma, boxsize=None, sed_min_step=0.0,
                 sed_rel_step=1e-2, center_step=3e-2, origin=None, center_rel_step=0.0,
                 psf_beta=0.0):
        self.center = np.array(center, dtype=np.float64).reshape(2)
        self.psf_sigma = float(psf_sigma)
        self.psf_beta = float(psf_beta)
        if boxsize is None:
            boxsize = int(np.ceil((5 if self.psf_beta > 0 else 10) * self.psf_sigma))
        if boxsize % 2 == 0:
            boxsize += 1
        if origin is None:
            pixel = np.round(self.center).astype(int)
            origin = (int(pixel[0]) - boxsize // 2, int(pixel[1]) - boxsize // 2)
        # else: the box of a source whose centre has already moved stays where it was
        super().__init__(sed, np.zeros((boxsize, boxsize), dtype=np.float32), origin,
                         sed_min_step=sed_min_step, sed_rel_step=sed_rel_step,
                         morph_step=center_step, morph_rel_step=center_rel_step,
                         prox_flags=_lib.COMPONENT_POINT_SOURCE)


class BlendBatch:
    """A batch of blends sharing the frame shape ``(C, H, W)``.

    Parameters
    ----------
    data, weights: (n_blends, C, H, W) float32 arrays
    components: list (one entry per blend) of lists of ``ComponentSpec``
    kernel: difference kernel, ``None`` (NullRenderer), ``(Ck, P, P)`` shared by
        all blends, or ``(n_blends, Ck, P, P)``; Ck in {1, C}
    max_iter: capacity of the loss history
    fft_shape: ``None`` or (Fy, Fx); any alias-free shape gives the same linear
        convolution as the reference's (fft.py:116-167)
    conv_path: "auto" (fused LDS-resident convolution kernel when the padded band
        fits the LDS, otherwise rocFFT), "rocfft" (rocFFT pipeline with the
        reference's FFT shape by default) or "fused"
    device: GPU index
    scheme: "amsgrad" (``proxmin.adaprox`` as used by ``Blend.fit`` and lite's
        ``AdaproxParameter``) or "fista" (lite's ``FistaParameter``; every component
        needs ``fista_step``)
    log_norm: include the normalisation term of ``Observation.log_norm`` in the loss
        (False = scarlet.lite's loss, lite/models.py:541)
    """

    def __init__(self, data, weights, components, kernel=None, max_iter=200,
                 fft_shape=None, device=0, conv_path="auto", scheme="amsgrad", log_norm=True):
        lib = _lib.load()
        self._lib = lib
        self._h = ctypes.c_void_p()
        data = _lib.f32(data)
        weights = _lib.f32(weights)
        assert data.ndim == 4 and data.shape == weights.shape
        nb, C, H, W = data.shape
        assert len(components) == nb
        self.n_blends, self.C, self.H, self.W = nb, C, H, W
        self.n_comp_per_blend = [len(c) for c in components]
        flat = [c for blend in components for c in blend]
        self.n_components = len(flat)
        self.max_iter = int(max_iter)

        desc = _lib.BatchDesc()
        desc.n_blends, desc.C, desc.H, desc.W = nb, C, H, W
        desc.n_components = len(flat)
        desc.max_iter = self.max_iter
        if kernel is not None:
            kernel = _lib.f32(kernel)
            per_blend = kernel.ndim == 4
            kb = kernel.shape[-3]
            assert kb in (1, C)
            if per_blend:
                assert kernel.shape[0] == nb
            desc.kernel_h, desc.kernel_w = kernel.shape[-2:]
            desc.kernel_bands, desc.kernel_per_blend = kb, int(per_blend)
        if fft_shape is not None:
            desc.fft_h, desc.fft_w = int(fft_shape[0]), int(fft_shape[1])
        desc.conv_path = {"auto": 0, "rocfft": 1, "fused": 2}[conv_path]
        _lib.check(lib.smi_batch_create(ctypes.byref(desc), int(device), ctypes.byref(self._h)))
        self.scheme = scheme
        _lib.check(lib.smi_batch_set_scheme(
            self._h, {"amsgrad": _lib.SCHEME_AMSGRAD, "fista": _lib.SCHEME_FISTA}[scheme]))
        _lib.check(lib.smi_batch_set_log_norm(self._h, int(bool(log_norm))))

        self._plan_ids = {}
        comps = self._pack_components(flat)
        _lib.check(lib.smi_batch_set_components(self._h, ctypes.byref(comps)))
        _lib.check(
            lib.smi_batch_set_observation(
                self._h, _lib.ptr(data, ctypes.c_float), _lib.ptr(weights, ctypes.c_float)
            )
        )
        self._kernel_shape = None if kernel is None else kernel.shape
        if kernel is not None:
            _lib.check(lib.smi_batch_set_kernel(self._h, _lib.ptr(kernel, ctypes.c_float)))

    # -- component tables ---------------------------------------------------
    def _pack_components(self, flat, values=True, rows=None):
        """``smi_components`` of the ComponentSpecs ``flat`` (all blends, in order); registers
        the monotonicity plans their boxes need.  ``values=False``: ``sed`` and ``morph`` are
        left unset (``smi_batch_update_components`` does not read them).  ``rows``: indices of
        the components that differ from the last call -- only their rows are written again
        (the per-component loops below are what a thousand-blend ``fit_blends`` would
        otherwise spend its hook rounds in)."""
        lib, nb, C = self._lib, self.n_blends, self.C
        # monotonicity plans, one per (box shape, weighting); with centre fitting
        # (PROX_FIT_CENTER) nine consecutive ones for the centres around the box centre
        plan_ids = self._plan_ids
        part = flat if rows is None else [flat[k] for k in rows]
        for c in part:
            if c.prox_flags & _lib.PROX_MONOTONIC:
                self._plan_for(c.morph.shape, c.neighbor_weight,
                               bool(c.prox_flags & _lib.PROX_FIT_CENTER))

        shapes = [c.morph.shape for c in part]
        fields = dict(
            origin_y=_lib.i32([c.origin[0] for c in part]),
            origin_x=_lib.i32([c.origin[1] for c in part]),
            box_h=_lib.i32([s[0] for s in shapes]),
            box_w=_lib.i32([s[1] for s in shapes]),
            sed_min_step=_lib.f32(
                np.stack([c.sed_min_step for c in part]) if part else np.zeros((0, C))
            ),
            sed_rel_step=_lib.f32([c.sed_rel_step for c in part]),
            morph_step=_lib.f32([c.morph_step for c in part]),
            prox_flags=_lib.i32([c.prox_flags for c in part]),
            sweep_plan=_lib.i32(
                [
                    plan_ids.get((c.morph.shape, c.neighbor_weight,
                                  bool(c.prox_flags & _lib.PROX_FIT_CENTER)), -1)
                    if c.prox_flags & _lib.PROX_MONOTONIC else -1
                    for c in part
                ]
            ),
            min_gradient=_lib.f32([c.min_gradient for c in part]),
            l_thresh=_lib.f32([c.l_thresh for c in part]),
            morph_rel_step=_lib.f32([c.morph_rel_step for c in part]),
            center=np.ascontiguousarray(
                [getattr(c, "center", (0.0, 0.0)) for c in part], dtype=np.float64
            ).reshape(-1, 2),
            psf_sigma=_lib.f32([getattr(c, "psf_sigma", 0.0) for c in part]),
            shift_step=_lib.f32([c.shift_step for c in part]),
            center_floor=_lib.f32([c.center_floor for c in part]),
            bg_level=_lib.f32(
                np.stack([c.bg_level if c.bg_level is not None else np.zeros(C, np.float32)
                          for c in part]) if part else np.zeros((0, C))
            ),
            fista_step=_lib.f32([c.fista_step for c in part]),
            sym_strength=_lib.f32([c.sym_strength for c in part]),
            chain_repeat=np.ascontiguousarray([c.chain_repeat for c in part], dtype=np.int32),
            pos_floor=_lib.f32([c.pos_floor for c in part]),
            shift_rel_step=_lib.f32([c.shift_rel_step for c in part]),
            psf_beta=_lib.f32([getattr(c, "psf_beta", 0.0) for c in part]),
        )
        if rows is None:
            arrays = fields
            arrays["blend"] = _lib.i32(np.repeat(np.arange(nb), self.n_comp_per_blend))
            self._shapes = shapes
            self._flags = [c.prox_flags for c in flat]
        else:
            arrays = self._component_arrays
            for name, sub in fields.items():
                arrays[name][rows] = sub
            for k, c in zip(rows, part):
                self._shapes[k] = c.morph.shape
                self._flags[k] = c.prox_flags
        self._component_arrays = arrays
        self._morph_offsets = np.concatenate(
            [[0], np.cumsum(arrays["box_h"].astype(np.int64) * arrays["box_w"])]
        ).astype(np.int64)
        arrays["sed"] = _lib.f32(np.stack([c.sed for c in flat]) if flat and values
                                 else np.zeros((0, C)))
        arrays["morph"] = _lib.f32(np.concatenate([c.morph.reshape(-1) for c in flat])
                                   if flat and values else np.zeros(0))
        comps = _lib.Components()
        for name, ctype in _lib.Components._fields_:
            setattr(comps, name, arrays[name].ctypes.data_as(ctype))
        comps._keepalive = arrays  # the ctypes struct only holds pointers
        return comps

    # -- box resizing on a live batch (include/scarlet_amd.h) ---------------------------
    def resize_test(self):
        """(margin, edge_pull) per component: the reductions ``ImageMorphology.update``
        decides on (morphology.py:132-207), computed on the device."""
        margin = np.zeros(self.n_components, dtype=np.int32)
        pull = np.zeros(self.n_components, dtype=np.float64)
        _lib.check(self._lib.smi_batch_resize_test(
            self._h, _lib.ptr(margin, ctypes.c_int32), _lib.ptr(pull, ctypes.c_double)))
        return margin, pull

    def component_states(self, indices):
        """Parameters and AMSGrad moments of the components ``indices``: one dict per
        component with ``sed, m_sed, v_sed, vhat_sed`` (C,) and ``morph, m_morph, v_morph,
        vhat_morph`` (h, w) float32 views into one download."""
        idx = _lib.i32(indices)
        C = self.C
        sizes = [4 * C + 4 * self._shapes[k][0] * self._shapes[k][1] for k in idx]
        buf = np.empty(int(np.sum(sizes)), dtype=np.float32)
        _lib.check(self._lib.smi_batch_get_component_states(
            self._h, _lib.ptr(idx, ctypes.c_int32), idx.size, _lib.ptr(buf, ctypes.c_float)))
        out, pos = [], 0
        for k, size in zip(idx, sizes):
            rec = buf[pos:pos + size]
            pos += size
            shape = self._shapes[k]
            n = shape[0] * shape[1]
            small = rec[:4 * C].reshape(4, C)
            px = rec[4 * C:].reshape(4, n)
            out.append(dict(sed=small[0], m_sed=small[1], v_sed=small[2], vhat_sed=small[3],
                            morph=px[0].reshape(shape), m_morph=px[1].reshape(shape),
                            v_morph=px[2].reshape(shape), vhat_morph=px[3].reshape(shape)))
        return out

    def update_components(self, components, keep, states, resized=None):
        """New component table on the live batch (after a box resize).  ``components``: the
        ComponentSpecs of all blends like at construction; ``keep`` (per component): 1 the
        device-resident parameters and moments stay (box unchanged); 0 they come from
        ``states``: in order, dicts like ``component_states`` returns (missing moments =
        zeros) with the arrays of the new box; 2 / 3 the box is resized about its centre on the
        device (include/scarlet_amd.h: centred slice, or zero-padded moments and a
        ``linear_ramp``-padded image -- 3 when the host image is float64) -- such a row takes
        its new table entries from ``resized``: ``rows`` (indices) with ``origin_y``,
        ``origin_x``, ``size``, ``morph_step`` per row; its ComponentSpec is not looked at
        except for the weighting of its monotonicity plan."""
        flat = [c for blend in components for c in blend]
        assert len(flat) == self.n_components and \
            [len(c) for c in components] == self.n_comp_per_blend
        keep = np.ascontiguousarray(keep, dtype=np.int32)
        if resized is not None and len(resized["rows"]):
            self._resize_rows(flat, **resized)
        comps = self._pack_components(flat, values=False, rows=np.flatnonzero(keep == 0))
        C = self.C
        parts = []
        it = iter(states)
        for k in np.flatnonzero(keep == 0):
            st = next(it)
            shape = self._shapes[k]
            zero_c, zero_n = np.zeros(C, np.float32), np.zeros(shape, np.float32)
            for name in ("sed", "m_sed", "v_sed", "vhat_sed"):
                a = st.get(name)
                parts.append(zero_c if a is None else np.asarray(a, dtype=np.float32).reshape(C))
            for name in ("morph", "m_morph", "v_morph", "vhat_morph"):
                a = st.get(name)
                a = zero_n if a is None else np.asarray(a, dtype=np.float32)
                assert a.shape == tuple(shape), (name, a.shape, shape)
                parts.append(a.reshape(-1))
        buf = np.concatenate(parts) if parts else np.zeros(1, np.float32)
        _lib.check(self._lib.smi_batch_update_components(
            self._h, ctypes.byref(comps), _lib.ptr(keep, ctypes.c_int32),
            _lib.ptr(_lib.f32(buf), ctypes.c_float)))

    def _plan_for(self, shape, weighting, fit_center):
        """Id of the monotonicity plan of a (box shape, weighting) pair, registered on first
        use; with centre fitting (PROX_FIT_CENTER) the first of nine consecutive plans for the
        centres around the box centre."""
        shape = (int(shape[0]), int(shape[1]))
        key = (shape, weighting, bool(fit_center))
        if key in self._plan_ids:
            return self._plan_ids[key]
        lib = self._lib

        def add_plan(center):
            wts, off, didx = operator.monotonic_tables(shape, weighting, center)
            return _lib.check(lib.smi_batch_add_sweep_plan(
                self._h, shape[0], shape[1], _lib.ptr(wts, ctypes.c_double),
                _lib.ptr(off, ctypes.c_int32), _lib.ptr(didx, ctypes.c_int32), didx.size))

        if not fit_center:
            self._plan_ids[key] = add_plan(None)
            return self._plan_ids[key]
        h, w = shape
        if h < 3 or w < 3:
            raise ValueError("centre fitting needs boxes of at least 3x3 pixels")
        ids = [add_plan((h // 2 + dy, w // 2 + dx)) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]
        if ids != list(range(ids[0], ids[0] + 9)):
            raise RuntimeError("the nine centre plans of a box must be consecutive")
        self._plan_ids[key] = ids[0]
        return ids[0]

    def _resize_rows(self, flat, rows, origin_y, origin_x, size, morph_step):
        """Table entries of rows whose square boxes the device resizes (``update_components``)."""
        arrays = self._component_arrays
        rows = np.asarray(rows, dtype=np.int64)
        size = np.asarray(size, dtype=np.int32)
        arrays["origin_y"][rows] = origin_y
        arrays["origin_x"][rows] = origin_x
        arrays["box_h"][rows] = size
        arrays["box_w"][rows] = size
        arrays["morph_step"][rows] = morph_step
        for k, n in zip(rows.tolist(), size.tolist()):
            c = flat[k]
            shape = (n, n)
            self._shapes[k] = shape
            if not c.prox_flags & _lib.PROX_MONOTONIC:
                continue
            # (centre fitting: the nine plans of the new shape, as _pack_components registers them)
            arrays["sweep_plan"][k] = self._plan_for(
                shape, c.neighbor_weight, bool(c.prox_flags & _lib.PROX_FIT_CENTER))

    def set_iteration_base(self, base):
        """Per blend: the iteration counter at which its current adaprox call began (``None``:
        0 for all); ``step(it0, n)`` then runs blends at different counters in one launch."""
        if base is None:
            _lib.check(self._lib.smi_batch_set_iteration_base(self._h, None))
            return
        base = np.ascontiguousarray(base, dtype=np.int32)
        assert base.shape == (self.n_blends,)
        _lib.check(self._lib.smi_batch_set_iteration_base(self._h, _lib.ptr(base, ctypes.c_int32)))

    def set_pause_at(self, it):
        """Per blend: the iteration counter (as passed to ``step``) after whose update the blend
        pauses for the rest of the call (``None``: nobody pauses); clears ``converged()``."""
        if it is None:
            _lib.check(self._lib.smi_batch_set_pause_at(self._h, None))
            return
        it = np.ascontiguousarray(it, dtype=np.int32)
        assert it.shape == (self.n_blends,)
        _lib.check(self._lib.smi_batch_set_pause_at(self._h, _lib.ptr(it, ctypes.c_int32)))

    def converged(self):
        """1 per blend whose stopping rule fired since the last ``set_pause_at``."""
        flag = np.zeros(self.n_blends, dtype=np.int32)
        _lib.check(self._lib.smi_batch_get_converged(self._h, _lib.ptr(flag, ctypes.c_int32)))
        return flag

    def set_round(self, states=None, base=None, pause_at=None):
        """``set_states`` + ``set_iteration_base`` + ``set_pause_at`` in one call (``None``:
        left as it is on the device)."""
        arrs = []
        for a in (states, base, pause_at):
            if a is not None:
                a = np.ascontiguousarray(a, dtype=np.int32)
                assert a.shape == (self.n_blends,)
            arrs.append(a)
        _lib.check(self._lib.smi_batch_set_round(
            self._h, *[_lib.ptr(a, ctypes.c_int32) for a in arrs]))

    def round(self):
        """``progress()`` and ``converged()`` in one call: (state, losses recorded, stopped by
        its own rule) per blend; blocks until the steps are done."""
        out = [np.zeros(self.n_blends, dtype=np.int32) for _ in range(3)]
        _lib.check(self._lib.smi_batch_get_round(
            self._h, *[_lib.ptr(a, ctypes.c_int32) for a in out]))
        return out

    def set_states(self, states):
        """Per-blend state: 0 iterating, 2 finished or paused (skipped by every kernel),
        3 failed."""
        st = np.ascontiguousarray(states, dtype=np.int32)
        assert st.shape == (self.n_blends,)
        _lib.check(self._lib.smi_batch_set_states(self._h, _lib.ptr(st, ctypes.c_int32)))

    def progress(self):
        """(state, number of recorded losses) per blend; blocks until the steps are done."""
        st = np.zeros(self.n_blends, dtype=np.int32)
        n = np.zeros(self.n_blends, dtype=np.int32)
        _lib.check(self._lib.smi_batch_get_progress(
            self._h, _lib.ptr(st, ctypes.c_int32), _lib.ptr(n, ctypes.c_int32)))
        return st, n

    # -- lifetime ----------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.smi_batch_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers -----------------------------------------------------------
    def _split_morphs(self, flat):
        off = self._morph_offsets.tolist()
        return [flat[off[k]:off[k + 1]].reshape(self._shapes[k]) for k in range(self.n_components)]

    @property
    def fft_shape(self):
        fy, fx = ctypes.c_int32(), ctypes.c_int32()
        _lib.check(self._lib.smi_batch_fft_shape(self._h, ctypes.byref(fy), ctypes.byref(fx)))
        return fy.value, fx.value

    @property
    def conv_path(self):
        """The convolution path this batch runs: "none" (NullRenderer), "rocfft" or "fused"."""
        path = ctypes.c_int32()
        _lib.check(self._lib.smi_batch_conv_path_used(self._h, ctypes.byref(path)))
        return ("none", "rocfft", "fused")[path.value]

    def set_kernel(self, kernel):
        """Replace the difference kernel (same stamp shape as at construction)."""
        kernel = _lib.f32(kernel)
        assert kernel.shape == self._kernel_shape, "kernel shape is fixed at construction"
        _lib.check(self._lib.smi_batch_set_kernel(self._h, _lib.ptr(kernel, ctypes.c_float)))

    def set_kernel_shift(self, kernel, shift, fft_shape, step=1e-2, m=None, v=None, vhat=None):
        """``ConvolutionRenderer(psf_shift=...)`` (renderer.py:175-177, 215-228): from now on
        the difference kernel is ``fft.shift(kernel, shift)`` with ``shift`` a free
        parameter of the fit (``smi_batch_set_kernel_shift``).

        kernel: the unshifted stamps ``(n_sets, kernel_bands, h0, w0)`` (or without the
            leading axis for one set), at most the stamp shape given at construction
        shift: ``(n_sets, 2)`` or ``(2,)``; ``m``, ``v``, ``vhat`` likewise (warm start)
        fft_shape: FFT lengths ``fft.shift`` uses for an ``(h0, w0)`` image"""
        kernel = _lib.f32(kernel)
        if kernel.ndim == 3:
            kernel = kernel[None]
        n_sets = kernel.shape[0]
        assert kernel.shape[1] == self._kernel_shape[-3], "number of kernel bands is fixed at construction"
        shift = np.ascontiguousarray(np.asarray(shift, dtype=np.float64).reshape(n_sets, 2))
        moments = None
        if m is not None or v is not None or vhat is not None:
            moments = np.ascontiguousarray(np.stack(
                [np.zeros((n_sets, 2)) if a is None else np.asarray(a, dtype=np.float64).reshape(n_sets, 2)
                 for a in (m, v, vhat)], axis=1))
        _lib.check(self._lib.smi_batch_set_kernel_shift(
            self._h, _lib.ptr(kernel, ctypes.c_float), kernel.shape[2], kernel.shape[3],
            _lib.ptr(_lib.i32(fft_shape), ctypes.c_int32), _lib.ptr(shift, ctypes.c_double),
            _lib.ptr(moments, ctypes.c_double), float(step)))
        self._kernel_sets = n_sets

    def set_kernel_shift_relative_step(self, factor):
        """``relative_step`` for the kernel shift (parameter.py:126-129): step =
        max(``step`` of :meth:`set_kernel_shift`, ``factor`` * mean(shift))."""
        _lib.check(self._lib.smi_batch_set_kernel_shift_relative_step(self._h, float(factor)))

    def kernel_shift(self, kernel=False):
        """State of the free kernel shift: dict of ``(n_sets, 2)`` float64 arrays ``shift``,
        ``m``, ``v``, ``vhat``, ``gradient`` (d(-logL)/d(shift) of the last ``step`` /
        ``gradient`` call) and, on request, ``kernel``: the stamps at the current shift."""
        n = self._kernel_sets
        shift, grad, mom = np.zeros((n, 2)), np.zeros((n, 2)), np.zeros((n, 3, 2))
        stamps = np.zeros((n,) + tuple(self._kernel_shape[-3:]), dtype=np.float32) if kernel else None
        _lib.check(self._lib.smi_batch_get_kernel_shift(
            self._h, _lib.ptr(shift, ctypes.c_double), _lib.ptr(mom, ctypes.c_double),
            _lib.ptr(grad, ctypes.c_double), _lib.ptr(stamps, ctypes.c_float)))
        out = dict(shift=shift, m=mom[:, 0].copy(), v=mom[:, 1].copy(), vhat=mom[:, 2].copy(),
                   gradient=grad)
        if kernel:
            out["kernel"] = stamps
        return out

    def set_optimizer(self, b1=0.9, b2=0.999, eps=1e-8):
        """AMSGrad constants (``proxmin.adaprox`` keywords b1, b2, eps)."""
        _lib.check(self._lib.smi_batch_set_optimizer(self._h, b1, b2, eps))

    def set_stream(self, stream_handle):
        """Launch on the given HIP stream (e.g. ``torch.cuda.current_stream().cuda_stream``)."""
        _lib.check(self._lib.smi_batch_set_stream(self._h, ctypes.c_void_p(stream_handle)))

    def set_observation(self, data, weights):
        """Replace data and weights (same shape; the normalisation term of the loss follows
        the new weights)."""
        data, weights = _lib.f32(data), _lib.f32(weights)
        assert data.shape == weights.shape == (self.n_blends, self.C, self.H, self.W)
        _lib.check(self._lib.smi_batch_set_observation(
            self._h, _lib.ptr(data, ctypes.c_float), _lib.ptr(weights, ctypes.c_float)))

    def set_observation_device(self, data_ptr, weights_ptr):
        """Adopt device-resident data/weights (e.g. ``tensor.data_ptr()``)."""
        _lib.check(
            self._lib.smi_batch_set_observation_device(
                self._h, ctypes.c_void_p(data_ptr), ctypes.c_void_p(weights_ptr)
            )
        )

    # -- forward / gradient --------------------------------------------------
    def forward(self, model=True, rendered=True):
        """``(model, rendered, logL)`` for every blend: Blend.get_model,
        Observation.render, Observation.get_log_likelihood."""
        shape = (self.n_blends, self.C, self.H, self.W)
        m = np.empty(shape, dtype=np.float32) if model else None
        r = np.empty(shape, dtype=np.float32) if rendered else None
        logL = np.empty(self.n_blends, dtype=np.float64)
        _lib.check(
            self._lib.smi_batch_forward(
                self._h, _lib.ptr(m, ctypes.c_float), _lib.ptr(r, ctypes.c_float),
                _lib.ptr(logL, ctypes.c_double),
            )
        )
        return m, r, logL

    def gradient(self):
        """Gradient of ``-logL``: (g_sed (n_components, C), list of g_morph)."""
        g_sed = np.empty((self.n_components, self.C), dtype=np.float32)
        g_morph = np.empty(int(self._morph_offsets[-1]), dtype=np.float32)
        _lib.check(
            self._lib.smi_batch_gradient(
                self._h, _lib.ptr(g_sed, ctypes.c_float), _lib.ptr(g_morph, ctypes.c_float)
            )
        )
        return g_sed, self._split_morphs(g_morph)

    # -- optimisation --------------------------------------------------------
    def step(self, it0, n_iter, e_rel=1e-3, min_iter=1, prox_max_iter=10,
             check_convergence=False):
        """``n_iter`` asynchronous iterations starting at counter ``it0``.
        ``e_rel`` is the tolerance of the proximal sub-iterations and, with
        ``check_convergence``, of the per-blend stopping rule (off by default:
        fixed-iteration runs)."""
        _lib.check(
            self._lib.smi_batch_step(
                self._h, it0, n_iter, e_rel, min_iter, prox_max_iter, int(check_convergence)
            )
        )

    def status(self):
        """(number of blends still iterating, first non-finite blend or -1); blocks."""
        a, e = ctypes.c_int32(), ctypes.c_int32()
        _lib.check(self._lib.smi_batch_status(self._h, ctypes.byref(a), ctypes.byref(e)))
        return a.value, e.value

    def states(self):
        """Per-blend state: 0 iterating, 1 last iteration, 2 converged, 3 non-finite."""
        out = np.zeros(self.n_blends, dtype=np.int32)
        _lib.check(self._lib.smi_batch_get_states(self._h, _lib.ptr(out, ctypes.c_int32)))
        return out

    def fit(self, max_iter=200, e_rel=1e-3, min_iter=1, prox_max_iter=10, sync_every=10):
        """Fit every blend; returns ``(n_iter, logL)`` arrays like the tuple
        ``Blend.fit`` returns (blend.py:194)."""
        if max_iter > self.max_iter:
            raise ValueError("max_iter exceeds the loss-history capacity of this batch")
        n_iter = np.zeros(self.n_blends, dtype=np.int32)
        _lib.check(
            self._lib.smi_batch_fit(
                self._h, max_iter, e_rel, min_iter, prox_max_iter, sync_every,
                _lib.ptr(n_iter, ctypes.c_int32),
            )
        )
        loss = self.loss_history()
        last = np.array([loss[b][n_iter[b] - 1] if n_iter[b] else np.nan
                         for b in range(self.n_blends)])
        return n_iter, -last

    def loss_history(self):
        """List (per blend) of the recorded losses (= -logL, blend.py:273)."""
        out = np.empty((self.n_blends, self.max_iter), dtype=np.float64)
        n = np.zeros(self.n_blends, dtype=np.int32)
        _lib.check(
            self._lib.smi_batch_get_loss(
                self._h, _lib.ptr(out, ctypes.c_double), self.max_iter,
                _lib.ptr(n, ctypes.c_int32),
            )
        )
        return [out[b, : min(n[b], self.max_iter)].copy() for b in range(self.n_blends)]

    def set_previous_loss(self, loss):
        """Seed the stopping rule's previous loss (a batch that continues another)."""
        loss = np.ascontiguousarray(np.broadcast_to(loss, (self.n_blends,)), dtype=np.float64)
        _lib.check(self._lib.smi_batch_set_previous_loss(self._h, _lib.ptr(loss, ctypes.c_double)))

    def add_observation(self, data, weights, kernel):
        """A further observation of every blend on the model's pixel grid (one more term
        of the loss and of the gradient image, blend.py:264-271): ``data`` / ``weights``
        (n_blends, C, H, W) over the model's channels, ``kernel`` with the stamp shape
        the batch was created with."""
        data, weights, kernel = _lib.f32(data), _lib.f32(weights), _lib.f32(kernel)
        assert data.shape == weights.shape == (self.n_blends, self.C, self.H, self.W)
        assert kernel.shape == self._kernel_shape, "kernel stamp shape is fixed at construction"
        _lib.check(self._lib.smi_batch_add_observation(
            self._h, _lib.ptr(data, ctypes.c_float), _lib.ptr(weights, ctypes.c_float),
            _lib.ptr(kernel, ctypes.c_float)))

    def add_loss_constant(self, constant):
        """Add a constant per blend to the loss (the part of an observation that lies
        outside the model frame)."""
        c = np.ascontiguousarray(np.broadcast_to(constant, (self.n_blends,)), dtype=np.float64)
        _lib.check(self._lib.smi_batch_add_loss_constant(self._h, _lib.ptr(c, ctypes.c_double)))

    def set_sub_ranges(self, n):
        """Split every step into ``n`` ranges of blends on streams of their own (0 =
        automatic).  Results do not depend on ``n``."""
        _lib.check(self._lib.smi_batch_set_sub_ranges(self._h, int(n)))

    def set_inline_render(self, on):
        """Plain batches (factorized components under one fused convolution): let the
        convolution kernel render its own model rows (default) or keep a model cube in HBM,
        written by the render kernel every iteration.  Same results bit for bit."""
        _lib.check(self._lib.smi_batch_set_inline_render(self._h, int(bool(on))))

    def sub_ranges(self):
        n = ctypes.c_int32()
        _lib.check(self._lib.smi_batch_get_sub_ranges(self._h, ctypes.byref(n)))
        return n.value

    def attach_lowres(self, resampler, channels, data, weights, log_norm):
        """Add a further observation of the (single) blend on a coarser pixel grid as a
        term of the loss and of the gradient (every call adds one).  ``resampler`` is the handle of a
        ``ResolutionRenderer``'s device operators (``smi_resampler``), ``channels`` the
        model channel of each of its bands, ``data`` / ``weights`` (C, n_a, n_b),
        ``log_norm`` the observation's ``log_norm``."""
        channels = np.ascontiguousarray(channels, dtype=np.int32)
        data = np.ascontiguousarray(data, dtype=np.float32)
        weights = np.ascontiguousarray(weights, dtype=np.float32)
        assert data.shape == weights.shape and data.shape[0] == len(channels)
        self._lowres_shapes = getattr(self, "_lowres_shapes", []) + [data.shape]
        _lib.check(self._lib.smi_batch_attach_lowres(
            self._h, resampler, _lib.ptr(channels, ctypes.c_int32),
            _lib.ptr(data, ctypes.c_float), _lib.ptr(weights, ctypes.c_float), float(log_norm)))

    def lowres_rendered(self, index=0):
        """Rendering of the ``index``-th attached observation in the last forward /
        gradient / step call."""
        out = np.empty(self._lowres_shapes[index], dtype=np.float32)
        _lib.check(self._lib.smi_batch_get_lowres_rendered(self._h, index,
                                                           _lib.ptr(out, ctypes.c_float)))
        return out

    def reset(self):
        _lib.check(self._lib.smi_batch_reset(self._h))

    def save_state(self):
        """Device-side copy of parameters and optimizer state (see ``restore_state``)."""
        _lib.check(self._lib.smi_batch_save_state(self._h))

    def restore_state(self):
        """Back to the saved parameters and optimizer state, blends re-armed, loss
        histories cleared; no host round trip."""
        _lib.check(self._lib.smi_batch_restore_state(self._h))

    # -- parameters and optimizer state ---------------------------------------
    def parameters(self):
        """(seds (n_components, C), list of morphologies)."""
        sed = np.empty((self.n_components, self.C), dtype=np.float32)
        morph = np.empty(int(self._morph_offsets[-1]), dtype=np.float32)
        _lib.check(
            self._lib.smi_batch_get_parameters(
                self._h, _lib.ptr(sed, ctypes.c_float), _lib.ptr(morph, ctypes.c_float)
            )
        )
        return sed, self._split_morphs(morph)

    def has_shift(self, k):
        """True if component ``k`` carries a Fourier shift on the device."""
        return bool(self._flags[k] & _lib.COMPONENT_SHIFTING)

    def model_morphologies(self):
        """The morphologies as they enter the model: the Fourier-shifted image of a
        shifting component, the PSF image of a point source, else the parameter."""
        morph = np.empty(int(self._morph_offsets[-1]), dtype=np.float32)
        _lib.check(self._lib.smi_batch_get_model_morphology(self._h, _lib.ptr(morph, ctypes.c_float)))
        return self._split_morphs(morph)

    def centers(self):
        """State of the free 2-vectors: dict of (n_components, 2) float64 arrays
        ``center`` (point sources: centre in frame pixels; shifting components: the
        shift; zeros otherwise), ``m``, ``v``, ``vhat`` and, after ``gradient()``,
        ``gradient``."""
        out = {k: np.zeros((self.n_components, 2)) for k in ("center", "m", "v", "vhat", "gradient")}
        _lib.check(
            self._lib.smi_batch_get_centers(
                self._h, *[_lib.ptr(out[k], ctypes.c_double)
                           for k in ("center", "m", "v", "vhat", "gradient")]
            )
        )
        return out

    def set_centers(self, center):
        """New point-source centres (frame pixels) / free shifts, (n_components, 2); rows of
        other components are ignored.  The morphologies that enter the model follow."""
        center = np.ascontiguousarray(center, dtype=np.float64).reshape(self.n_components, 2)
        _lib.check(self._lib.smi_batch_set_centers(self._h, _lib.ptr(center, ctypes.c_double)))

    def set_center_moments(self, m=None, v=None, vhat=None):
        arrs = [None if a is None else np.ascontiguousarray(a, dtype=np.float64).reshape(-1, 2)
                for a in (m, v, vhat)]
        _lib.check(
            self._lib.smi_batch_set_center_moments(
                self._h, *[_lib.ptr(a, ctypes.c_double) for a in arrs]
            )
        )

    def fista_state(self):
        """FISTA state: dict with ``z_sed`` (n_components, C), ``z_morph`` (list) and
        ``t`` (n_components, 2) for (spectrum, morphology)."""
        z_sed = np.empty((self.n_components, self.C), dtype=np.float32)
        z_morph = np.empty(int(self._morph_offsets[-1]), dtype=np.float32)
        t = np.empty((self.n_components, 2), dtype=np.float64)
        _lib.check(self._lib.smi_batch_get_fista_state(
            self._h, _lib.ptr(z_sed, ctypes.c_float), _lib.ptr(z_morph, ctypes.c_float),
            _lib.ptr(t, ctypes.c_double)))
        return dict(z_sed=z_sed, z_morph=self._split_morphs(z_morph), t=t)

    def set_fista_state(self, z_sed=None, z_morph=None, t=None):
        zs = None if z_sed is None else _lib.f32(z_sed)
        zm = None if z_morph is None else _lib.f32(
            np.concatenate([np.asarray(m).reshape(-1) for m in z_morph]))
        tt = None if t is None else np.ascontiguousarray(t, dtype=np.float64).reshape(-1, 2)
        _lib.check(self._lib.smi_batch_set_fista_state(
            self._h, _lib.ptr(zs, ctypes.c_float), _lib.ptr(zm, ctypes.c_float),
            _lib.ptr(tt, ctypes.c_double)))

    def set_parameters(self, seds=None, morphs=None):
        sed = None if seds is None else _lib.f32(seds)
        morph = None if morphs is None else _lib.f32(
            np.concatenate([np.asarray(m).reshape(-1) for m in morphs])
        )
        _lib.check(
            self._lib.smi_batch_set_parameters(
                self._h, _lib.ptr(sed, ctypes.c_float), _lib.ptr(morph, ctypes.c_float)
            )
        )

    def moments(self, dtype=np.float32):
        """AMSGrad state: dict with m/v/vhat for seds (arrays) and morphs (lists); with another
        ``dtype`` the six downloads are converted as a whole (the entries are views)."""
        ns, nm = (self.n_components, self.C), int(self._morph_offsets[-1])
        bufs = [np.empty(ns, dtype=np.float32) for _ in range(3)] + [
            np.empty(nm, dtype=np.float32) for _ in range(3)
        ]
        _lib.check(
            self._lib.smi_batch_get_moments(self._h, *[_lib.ptr(b, ctypes.c_float) for b in bufs])
        )
        if dtype != np.float32:
            bufs = [b.astype(dtype) for b in bufs]
        return dict(
            m_sed=bufs[0], v_sed=bufs[1], vhat_sed=bufs[2],
            m_morph=self._split_morphs(bufs[3]), v_morph=self._split_morphs(bufs[4]),
            vhat_morph=self._split_morphs(bufs[5]),
        )

    def set_moments(self, m_sed=None, v_sed=None, vhat_sed=None, m_morph=None, v_morph=None,
                    vhat_morph=None):
        def pack(x, is_morph):
            if x is None:
                return None
            if is_morph:
                return _lib.f32(np.concatenate([np.asarray(m).reshape(-1) for m in x]))
            return _lib.f32(x)

        arrs = [pack(m_sed, 0), pack(v_sed, 0), pack(vhat_sed, 0), pack(m_morph, 1),
                pack(v_morph, 1), pack(vhat_morph, 1)]
        _lib.check(
            self._lib.smi_batch_set_moments(self._h, *[_lib.ptr(a, ctypes.c_float) for a in arrs])
        )

    # -- timing ----------------------------------------------------------------
    def enable_timing(self, on=True):
        _lib.check(self._lib.smi_batch_enable_timing(self._h, int(on)))

    def timing(self):
        """Mean ms per iteration of (render, conv, residual, conv^T, update, total)
        of the last timed ``step`` call."""
        out = np.zeros(6, dtype=np.float64)
        _lib.check(self._lib.smi_batch_get_timing(self._h, _lib.ptr(out, ctypes.c_double), 6))
        return dict(zip(("render", "conv", "residual", "conv_adj", "update", "total"), out))
