"""``Blend``: the scene and its fit (reference scarlet/blend.py:49-308).

``fit`` keeps the reference's signature, return value and side effects (parameter
values updated in place, ``m / v / vhat / std`` on every ``Parameter``,
``blend.loss``), but the whole proximal-gradient loop -- what the reference
delegates to ``proxmin.adaprox`` with autograd gradients (blend.py:165-180) --
runs on the GPU through ``libscarlet_amd.so``.  The host only steps in every 10
iterations for the box-resizing hook (``src.update()``, blend.py:284-292).
"""

import logging
import os
from functools import partial

import numpy as np

from . import _lib, fft
from .batch import BlendBatch, ComponentSpec, PointSourceSpec
from .bbox import overlapped_slices
from .component import CombinedComponent, FactorizedComponent
from .constraint import PositivityConstraint, device_flags
from .hoststep import HostBandSource, HostParameter, HostVector
from .model import UpdateException
from .morphology import PointSourceMorphology
from .psf import GaussianPSF, ImagePSF, MoffatPSF
from .parameter import relative_step, STD_FROM_V
from .renderer import ConvolutionRenderer, NullRenderer, ResolutionRenderer

logger = logging.getLogger("scarlet_amd.blend")


class _HostSteppedShift(Exception):
    """the batch cannot step the kernel shift on the device (no fused convolution)"""


def _band_dependent_psf(comp):
    """True for a point source whose model PSF is an ImagePSF that differs between the bands."""
    morphology = comp.children[1]
    if not isinstance(morphology, PointSourceMorphology) or not isinstance(morphology.psf, ImagePSF):
        return False
    cube = np.asarray(morphology.psf.get_parameter(0))
    return cube.ndim == 3 and cube.shape[0] > 1 and not np.all(cube == cube[0])


def _band_components(src):
    """The device's stand-ins for a point source on a band-dependent ImagePSF
    (``hoststep.HostBandSource``): per band a factorized component -- that band's spectrum
    entry, the others zero, times that band's stamp under a free Fourier shift -- whose
    Parameters are fixed as far as the device is concerned.  Built once per source."""
    cached = getattr(src, "_band_stand_ins", None)
    if cached is not None:
        return cached
    from .morphology import ImageMorphology
    from .parameter import Parameter
    from .spectrum import TabulatedSpectrum

    spectrum, morphology = src.children
    cube = np.asarray(morphology.psf.get_parameter(0), dtype=np.float64)
    sed, center = spectrum._parameters[0], morphology._parameters[0]
    assert cube.shape[0] == sed.shape[0], "one stamp of the model PSF per band of the spectrum"
    box = morphology.bbox[-2:]
    middle = np.array(box.origin, dtype=np.float64) + np.array(box.shape) / 2
    out = []
    for c in range(cube.shape[0]):
        values = np.zeros(sed.shape, dtype=np.float64)
        values[c] = sed[c]
        comp = FactorizedComponent(
            src.frame,
            TabulatedSpectrum(src.frame, Parameter(values, name="spectrum", fixed=True),
                              bbox=spectrum.bbox),
            ImageMorphology(src.frame, Parameter(cube[c].copy(), name="image", fixed=True),
                            bbox=box.copy(), shifting=True,
                            shift=Parameter(np.asarray(center, dtype=np.float64) - middle,
                                            name="shift", step=0.0),
                            resizing=False))
        comp._band_of = (src, c, middle)
        out.append(comp)
    src._band_stand_ins = out
    return out


def _flatten(sources):
    """FactorizedComponents of the scene in parameter order (a point source on a band-dependent
    ImagePSF as its per-band stand-ins, ``_band_components``)."""
    out = []
    for src in sources:
        if isinstance(src, FactorizedComponent):
            # (the rare case is told apart by one type check per component)
            if isinstance(src._children[1], PointSourceMorphology) and _band_dependent_psf(src):
                out.extend(_band_components(src))
            else:
                out.append(src)
        elif isinstance(src, CombinedComponent):
            if src.operation != "add":
                # (the reference's own 'multiply' model is identically zero: its accumulator starts
                # at zeros and is multiplied into, component.py:262-275 -- reproduced by
                # CombinedComponent.get_model here; there is nothing to fit)
                raise NotImplementedError("CombinedComponent('multiply') cannot be fitted: its "
                                          "model is identically zero in the reference")
            out.extend(_flatten(src.children))
        else:
            raise NotImplementedError(
                "{} cannot be fitted on the device (only factorized components)".format(
                    type(src).__name__
                )
            )
    return out


def _adaprox_options(alg_kwargs):
    """``(prox_max_iter, {b1, b2, eps})`` of ``proxmin.adaprox``'s keyword arguments as
    ``Blend.fit`` passes them on (blend.py:165-180); ``scheme``, ``p`` and ``callback`` are the
    caller's to pop.  Anything else is an option of proxmin this loop does not have."""
    prox_max_iter = alg_kwargs.pop("prox_max_iter", 10)
    opt = dict(b1=alg_kwargs.pop("b1", 0.9), b2=alg_kwargs.pop("b2", 0.999),
               eps=alg_kwargs.pop("eps", 1e-8))
    if alg_kwargs:
        raise NotImplementedError("unsupported adaprox options: {}".format(sorted(alg_kwargs)))
    return prox_max_iter, opt


def _vector_rule(p, what):
    """``(on_device, rule)`` of a free 2-vector (shift, point-source centre, psf_shift).  The
    device takes the bare AMSGrad step of the reference's defaults (morphology.py:673-676,
    source.py:115, renderer.py:175-177); a vector with a prior, a constraint or a step callable
    of the user's is stepped on the host from the device's gradient (hoststep.HostVector), like
    any other parameter the device cannot express (blend.py:120-145).  ``rule`` = (constant,
    relative factor, minimum), or the callable."""
    if p.fixed and p.step is None:
        return True, (0.0, 0.0, 0.0)
    rule = _step_rule(p.step, what)
    if rule is None:
        return False, p.step
    return p.prior is None and p.constraint is None, rule


def _rule(p, what):
    """step rule of a parameter, or the callable itself if it is user code"""
    if p.fixed and p.step is None:
        return (0.0, 0.0, 0.0)  # never used (blend.py:107-115)
    rule = _step_rule(p.step, what)
    return p.step if rule is None else rule


def _step_rule(step, what):
    """(constant, relative factor, minimum) of a Parameter.step, or None for a rule the device
    does not have (a step callable of the user's, ``relative_step`` along an axis of an
    image): that parameter is stepped on the host."""
    if isinstance(step, partial) and step.func is relative_step:
        kw = step.keywords
        axis = kw.get("axis")
        # (the mean of a spectrum along its only axis is its mean: parameter.py:126-129)
        along_all = axis is None or (what == "spectrum" and axis in (0, -1, (0,), (-1,)))
        if not along_all or step.args:
            return None
        return 0.0, float(kw.get("factor", 0.1)), kw.get("minimum", 0)
    if step is relative_step:
        return 0.0, 0.1, 0
    if callable(step):
        return None
    return float(step), 0.0, 0


class Blend(CombinedComponent):
    """Collection of sources fitted jointly to one observation."""

    def __init__(self, sources, observations):
        self.sources = sources if hasattr(sources, "__iter__") else (sources,)
        self.observations = observations if hasattr(observations, "__iter__") else (observations,)
        super().__init__(self.sources)
        self.loss = []
        self.device = 0  # GPU the fit runs on

    # ------------------------------------------------------------------ device
    def _observation(self):
        """(data, weights, kernel) of the scene as ONE cube over the model channels.

        Several observations on the model's pixel grid (e.g. different instruments
        covering different channels) are merged: per-channel difference kernels are
        zero-padded to a common stamp, a NullRenderer observation contributes a delta
        kernel.  The summed log-likelihood of the reference's loop over observations
        (blend.py:265-271) is the log-likelihood of the merged cube.  Observations that
        share a model channel with an earlier one cannot be merged: they go into further
        cubes of the same layout, ``self._extra_layers`` (one more term of the loss and
        of the gradient each, ``smi_batch_add_observation``)."""
        C = self.frame.C
        spatial = tuple(self.frame.shape[1:])
        channels = list(self.frame.channels)
        layers = []  # dict(data, weights, kernels[C], taken[C])
        self._lowres = []
        self._loss_constant = 0.0  # observed pixels outside the model frame
        if len(self.observations) == 1:
            # the common case -- one observation on the model's own grid and channels -- needs
            # no merging: its arrays are the cube (no copies: a thousand blends are 0.7 GB)
            obs = self.observations[0]
            r = obs.renderer
            if (type(r) in (NullRenderer, ConvolutionRenderer) and not obs.parameters
                    and tuple(obs.shape) == tuple(self.frame.shape)
                    and list(obs.channels) == channels
                    and type(obs.data) is np.ndarray and obs.data.dtype == np.float32
                    and type(obs.weights) is np.ndarray and obs.weights.dtype == np.float32):
                self._extra_layers = []
                if type(r) is NullRenderer:
                    return obs.data, obs.weights, None
                k = np.asarray(r.kernel_image(), dtype=np.float32)
                if k.shape[-2] % 2 == 1 and k.shape[-1] % 2 == 1 and k.shape[0] in (1, C):
                    if k.shape[0] > 1 and all(np.array_equal(k[0], k[c]) for c in range(1, C)):
                        k = k[:1]
                    return obs.data, obs.weights, k

        def layer_for(idx):
            for layer in layers:
                if not layer["taken"][idx].any():
                    return layer
            layers.append(dict(data=np.zeros(self.frame.shape, dtype=np.float32),
                               weights=np.zeros(self.frame.shape, dtype=np.float32),
                               kernels=[None] * C, taken=np.zeros(C, dtype=bool)))
            return layers[-1]

        # (the observation whose kernel carries a free psf_shift becomes the first layer: that
        # is the one the device moves, smi_batch_set_kernel_shift)
        for obs in sorted(self.observations, key=lambda o: all(p.fixed for p in o.parameters)):
            r = obs.renderer
            idx = [channels.index(c) for c in obs.channels]
            if type(r) is ResolutionRenderer:
                # coarser pixel grid: its own term of the loss (smi_batch_attach_lowres)
                self._lowres.append((obs, idx))
                continue
            if type(r) not in (NullRenderer, ConvolutionRenderer):
                raise NotImplementedError(
                    "renderer {} cannot run on the device; a linear user-defined renderer "
                    "with an `adjoint` method is fitted through Blend.fit (host-rendered "
                    "mode)".format(type(r).__name__))
            if any(not p.fixed for p in obs.parameters) and getattr(self, "_psf", None) is None:
                raise NotImplementedError("free renderer parameters with several observations")
            layer = layer_for(idx)
            layer["taken"][idx] = True
            data, weights = layer["data"], layer["weights"]
            if tuple(obs.shape[1:]) == spatial:
                data[idx] = obs.data
                weights[idx] = obs.weights
            else:
                # the observation covers a part of the frame (renderer.py:130-161,
                # match_shape): zero weight everywhere else
                data_sl, model_sl = r.slices
                if tuple(obs.data[data_sl].shape[1:]) != tuple(
                        data[(slice(None),) + tuple(model_sl[1:])].shape[1:]):
                    raise NotImplementedError("observation and model frame overlap differently")
                for j, c in enumerate(idx):
                    data[c][tuple(model_sl[1:])] = obs.data[j][tuple(data_sl[1:])]
                    weights[c][tuple(model_sl[1:])] = obs.weights[j][tuple(data_sl[1:])]
                # where the observation sticks out of the frame the reference's model is
                # zero (match_shape): a constant sum w d^2 / 2 + that region's log_norm
                outside = np.ones(obs.data.shape, dtype=bool)
                outside[(slice(None),) + tuple(data_sl[1:])] = False
                w_out = np.where(outside, np.asarray(obs.weights, dtype=np.float64), 0.0)
                seen = w_out != 0
                if seen.any():
                    d_out = np.asarray(obs.data, dtype=np.float64)
                    self._loss_constant += float(
                        0.5 * np.sum(w_out * d_out * d_out)
                        + seen.sum() / 2 * np.log(2 * np.pi) - 0.5 * np.sum(np.log(w_out[seen])))
            if isinstance(r, ConvolutionRenderer):
                k = np.asarray(r.kernel_image(), dtype=np.float32)
                for j, c in enumerate(idx):
                    layer["kernels"][c] = k[j if k.shape[0] > 1 else 0]
        if not layers:
            layers.append(dict(data=np.zeros(self.frame.shape, dtype=np.float32),
                               weights=np.zeros(self.frame.shape, dtype=np.float32),
                               kernels=[None] * C, taken=np.zeros(C, dtype=bool)))
        stamps = [k for layer in layers for k in layer["kernels"] if k is not None]
        self._extra_layers = []
        if not stamps:
            if len(layers) > 1:
                raise NotImplementedError("several NullRenderer observations of one channel")
            return layers[0]["data"], layers[0]["weights"], None
        ph = max(k.shape[0] for k in stamps) | 1  # odd, so that the stamps share their centre
        pw = max(k.shape[1] for k in stamps) | 1

        def stamp_cube(kernels):
            kernel = np.zeros((C, ph, pw), dtype=np.float32)
            for c, k in enumerate(kernels):
                if k is None:
                    kernel[c, ph // 2, pw // 2] = 1
                else:
                    oy, ox = ph // 2 - k.shape[0] // 2, pw // 2 - k.shape[1] // 2
                    kernel[c, oy:oy + k.shape[0], ox:ox + k.shape[1]] = k
            return kernel

        cubes = [stamp_cube(layer["kernels"]) for layer in layers]
        if len(layers) == 1 and all(np.array_equal(cubes[0][0], cubes[0][c]) for c in range(1, C)):
            cubes[0] = cubes[0][:1]
        self._extra_layers = [(layer["data"], layer["weights"], cube)
                              for layer, cube in zip(layers[1:], cubes[1:])]
        return layers[0]["data"], layers[0]["weights"], cubes[0]

    def _specs(self, comps):
        """Device description of every component.  Parameters whose constraint chain or
        step rule the device cannot express (user ``Constraint`` subclasses, built-in
        chains in another order, custom step callables) are listed in
        ``self._host`` as ``(component index, HostParameter)``: the device treats them
        as fixed and without constraint, the host updates them (hoststep.py)."""
        specs = []
        self._host = []
        for k, comp in enumerate(comps):
            spectrum, morphology = comp.children
            sed = spectrum._parameters[0]
            image = morphology._parameters[0]
            if getattr(comp, "_band_of", None) is not None:
                # stand-in of a point source on a band-dependent ImagePSF: fixed on the device,
                # the source's spectrum and centre are the host's (HostBandSource)
                src, c, middle = comp._band_of
                real_sed, real_center = src.children[0]._parameters[0], src.children[1]._parameters[0]
                if self._scheme_args()[0] != "amsgrad":
                    raise NotImplementedError("point sources with another scheme than amsgrad")
                values = np.zeros(real_sed.shape)
                values[c] = real_sed[c]
                sed[...] = values
                shift = morphology._parameters[1]
                shift[...] = np.asarray(real_center, dtype=np.float64) - middle
                if c == 0:
                    _, center_rule = _vector_rule(real_center, "center")
                    self._host.append((k, HostBandSource(
                        real_sed, real_center, _rule(real_sed, "spectrum"), center_rule, middle,
                        real_sed.shape[0])))
                specs.append(ComponentSpec(
                    np.asarray(sed), np.asarray(image), morphology.bbox.origin[-2:],
                    sed_min_step=0.0, sed_rel_step=0.0, morph_step=0.0,
                    prox_flags=_lib.COMPONENT_FIXED_MORPH | _lib.COMPONENT_FIXED_SED,
                    shift=np.asarray(shift, dtype=np.float64), shift_step=0.0))
                continue
            if isinstance(morphology, PointSourceMorphology):
                if self._scheme_args()[0] != "amsgrad":
                    raise NotImplementedError("point sources with another scheme than amsgrad")
                on_device, rule = _vector_rule(image, "center")
                if on_device and rule[1] and isinstance(morphology.psf, ImagePSF):
                    # (a stamp with a free Fourier shift on the device holds the OFFSET of the
                    # centre: a relative_step rule on the centre itself is the host's)
                    on_device = False
                if not on_device:  # a prior / constraint / step callable on the centre
                    self._host.append((k, HostVector(image, rule)))
                    rule = (0.0, 0.0, 0.0)
                # the spectrum like any other spectrum: PositivityConstraint(1e-20) and a
                # built-in step on the device, anything else (a prior, another constraint, a
                # step callable) stepped on the host from the device's gradient
                sed_rule = _rule(sed, "spectrum")
                free_form = isinstance(sed.constraint, PositivityConstraint) and sed.constraint.zero == 1e-20
                sed_on_device = not callable(sed_rule) and sed.prior is None and (
                    free_form or (sed.fixed and sed.constraint is None and np.all(np.asarray(sed) > 1e-20)))
                if not sed_on_device:
                    self._host.append((k, HostParameter(sed, "sed", sed_rule, *self._scheme_args())))
                    sed_rule = (0.0, 0.0, 0.0)
                specs.append(self._point_spec(sed, image, morphology, rule, sed_rule, sed_on_device))
                continue
            shift_kw = {}
            if getattr(morphology, "shifting", False):
                # Fourier sub-pixel shift (morphology.py:124-130).  A fixed zero shift --
                # the reference's default for a bare ImageMorphology, morphology.py:113 --
                # is the identity; a fixed non-zero one is applied with step 0
                shift = morphology._parameters[1]
                if not shift.fixed or np.any(np.asarray(shift) != 0):
                    # (relative_step, parameter.py:126-129: max(minimum, factor * mean))
                    on_device, rule = _vector_rule(shift, "shift")
                    if not on_device:  # the host steps it: the device keeps it where it is
                        self._host.append((k, HostVector(shift, rule)))
                        rule = (0.0, 0.0, 0.0)
                    const, rel, low = rule
                    const = max(const, float(np.max(low)))
                    if max(image.shape) > 240:
                        raise NotImplementedError(
                            "a component with a free Fourier shift is limited to boxes of "
                            "240 pixels a side on the device (got {})".format(image.shape))
                    shift_kw = dict(shift=np.asarray(shift), shift_step=0.0 if shift.fixed else const,
                                    shift_rel_step=0.0 if shift.fixed else rel)
            if shift_kw and (sed.prior is not None or image.prior is not None):
                raise NotImplementedError("priors on a component with a free Fourier shift")

            sed_rule, morph_rule = _rule(sed, "spectrum"), _rule(image, "morphology")
            # the spectrum kernel applies PositivityConstraint(1e-20) (spectrum.py:54-56)
            free_form = isinstance(sed.constraint, PositivityConstraint) and sed.constraint.zero == 1e-20
            # a prior is user code: its gradient joins the likelihood's on the host
            # (blend.py:120-131), so the parameter is stepped there
            device_scheme = getattr(self, "_scheme", ("amsgrad", 0.25))[0] == "amsgrad"
            if shift_kw and not device_scheme:
                raise NotImplementedError("free Fourier shifts with another scheme than amsgrad")
            sed_on_device = device_scheme and not callable(sed_rule) and sed.prior is None and (
                free_form or (sed.fixed and sed.constraint is None and np.all(np.asarray(sed) > 1e-20)))
            try:
                flags = device_flags(image.constraint)
                morph_on_device = device_scheme and not callable(morph_rule) and image.prior is None
            except NotImplementedError:
                flags = device_flags(None)
                morph_on_device = False
            if not (sed_on_device and morph_on_device) and shift_kw:
                raise NotImplementedError(
                    "user-defined constraints / steps on a component with a free Fourier shift")
            if not sed_on_device:
                self._host.append((k, HostParameter(sed, "sed", sed_rule, *self._scheme_args())))
                sed_rule = (0.0, 0.0, 0.0)
            if not morph_on_device:
                self._host.append((k, HostParameter(image, "morph", morph_rule, *self._scheme_args())))
                morph_rule = (0.0, 0.0, 0.0)
            s_const, s_rel, s_min = sed_rule
            m_const, m_rel, m_min = morph_rule
            flags["flags"] |= (_lib.COMPONENT_FIXED_SED if sed.fixed or not sed_on_device else 0) | (
                _lib.COMPONENT_FIXED_MORPH if image.fixed or not morph_on_device else 0)
            specs.append(
                ComponentSpec(
                    np.asarray(sed), np.asarray(image), morphology.bbox.origin[-2:],
                    sed_min_step=np.maximum(np.asarray(s_min, dtype=np.float64), s_const),
                    sed_rel_step=s_rel,
                    morph_step=max(m_const, m_min if isinstance(m_min, (int, float))
                                   else float(np.max(m_min))),
                    morph_rel_step=m_rel,
                    prox_flags=flags["flags"],
                    neighbor_weight=flags["neighbor_weight"] or "angle",
                    min_gradient=flags["min_gradient"],
                    l_thresh=flags["l_thresh"],
                    center_floor=flags["center_floor"],
                    sym_strength=flags["sym_strength"],
                    chain_repeat=flags["chain_repeat"],
                    pos_floor=flags["zero"],
                    **shift_kw,
                )
            )
        return specs

    def _scheme_args(self):
        return getattr(self, "_scheme", ("amsgrad", 0.25))

    def _host_gradients(self, batch, comps):
        """(g_sed, g_morph, g_vec) at the parameters of this iteration for the parameters the
        host steps; g_vec = gradients of the free 2-vectors when one of them is the host's."""
        g_sed, g_morph = batch.gradient()
        g_vec = None
        if any(hp.kind in ("vec", "band") for _, hp in self._host):
            g_vec = batch.centers()["gradient"]
            # a point source on an ImagePSF is a stamp with a free shift on the device: the
            # device holds the offset of the centre from the middle of its box (_point_spec)
            self._vec_offset = {}
            for k, hp in self._host:
                if hp.kind != "vec":
                    continue
                morphology = comps[k].children[1]
                if isinstance(morphology, PointSourceMorphology) and batch.has_shift(k):
                    bbox = morphology.bbox
                    self._vec_offset[k] = np.array(bbox.origin[-2:]) + np.array(bbox.shape[-2:]) / 2
                else:
                    self._vec_offset[k] = 0.0
        return g_sed, g_morph, g_vec

    def _host_update(self, batch, local, grads, e_rel, prox_max_iter, opt):
        """The host's share of iteration ``local``: AMSGrad + proximal sub-iterations of
        the parameters in ``self._host`` from the gradients the device gathered before its
        own update, then the new values go back to the device."""
        g_sed, g_morph, g_vec = grads
        seds, morphs = batch.parameters()
        centers = None
        for k, hp in self._host:
            if hp.kind == "sed":
                seds[k] = hp.update(local, g_sed[k], e_rel, prox_max_iter, **opt)
            elif hp.kind == "vec":
                # a free 2-vector (shift / centre) with a prior, a constraint or a step callable
                if centers is None:
                    centers = batch.centers()["center"]
                centers[k] = hp.update(local, g_vec[k], e_rel, prox_max_iter, **opt) - self._vec_offset[k]
            elif hp.kind == "band":
                # a point source on a band-dependent ImagePSF: its C stand-ins k .. k + C - 1
                if centers is None:
                    centers = batch.centers()["center"]
                rows = range(k, k + hp.n_bands)
                new_sed, offset = hp.update(local, [g_sed[j] for j in rows], [g_vec[j] for j in rows],
                                            e_rel, prox_max_iter, **opt)
                for c, j in enumerate(rows):
                    seds[j] = np.zeros_like(seds[j])
                    seds[j][c] = new_sed[c]
                    centers[j] = offset
            else:
                morphs[k] = hp.update(local, g_morph[k], e_rel, prox_max_iter, **opt)
            hp.store()
        if any(hp.kind != "vec" for _, hp in self._host):
            batch.set_parameters(seds, morphs)
        if centers is not None:
            batch.set_centers(centers)

    @staticmethod
    def _point_spec(sed, center, morphology, center_rule, sed_rule, sed_on_device=True):
        """PointSource -> device description; the model PSF must be a pixel-integrated
        Gaussian or a Moffat profile, the same in all bands (what the device kernel
        evaluates).  ``center_rule``: (constant, relative factor, minimum) of the centre's
        step on the device -- zeros for a centre the host steps (``_vector_rule``); ``sed_rule``
        likewise for the spectrum (``sed_on_device`` False: the host steps it)."""
        psf = morphology.psf
        moffat = isinstance(psf, MoffatPSF) and psf.is_same and \
            bool(np.all(psf.get_parameter(1) == psf.get_parameter(1)[0]))
        stamp = None
        if isinstance(psf, ImagePSF):
            cube = np.asarray(psf.get_parameter(0))
            if np.all(cube == cube[0]):
                stamp = cube[0]
        if stamp is None and not (
                moffat or (isinstance(psf, GaussianPSF) and psf.integrate and psf.is_same)):
            raise NotImplementedError(
                "point sources need a pixel-integrated GaussianPSF, a MoffatPSF or an ImagePSF "
                "model PSF, the same in all bands")
        s_const, s_rel, s_min = sed_rule
        sed_fixed = sed.fixed or not sed_on_device
        c_const, c_rel, c_low = center_rule
        c_const = max(c_const, float(np.max(c_low)))  # (relative_step: max(minimum, factor * mean))
        if stamp is not None:
            # ImagePSF.get_model(offset) is fft.shift of the stored image (psf.py:228-234): on
            # the device a component with a fixed image and a free Fourier shift -- the offset of
            # the centre from the mean of the box bounds (morphology.py:503-507)
            origin = morphology.bbox.origin[-2:]
            box_center = np.array(origin, dtype=np.float64) + np.array(stamp.shape) / 2
            return ComponentSpec(
                np.asarray(sed), stamp, origin,
                sed_min_step=np.maximum(np.asarray(s_min, dtype=np.float64), s_const),
                sed_rel_step=s_rel, morph_step=0.0,
                prox_flags=_lib.COMPONENT_FIXED_MORPH | (_lib.COMPONENT_FIXED_SED if sed_fixed else 0),
                shift=np.asarray(center, dtype=np.float64) - box_center,
                shift_step=0.0 if center.fixed else c_const)
        spec = PointSourceSpec(
            np.asarray(sed), np.asarray(center), float(psf.get_parameter(0)[0]),
            boxsize=morphology.bbox.shape[-1],
            sed_min_step=np.maximum(np.asarray(s_min, dtype=np.float64), s_const),
            sed_rel_step=s_rel, center_step=c_const, center_rel_step=c_rel,
            origin=morphology.bbox.origin[-2:],
            psf_beta=float(psf.get_parameter(1)[0]) if moffat else 0.0)
        # Parameter(fixed=True): zero gradient for that parameter (blend.py:107-115)
        spec.prox_flags |= (_lib.COMPONENT_FIXED_SED if sed_fixed else 0) | (
            _lib.COMPONENT_FIXED_MORPH if center.fixed else 0)
        return spec

    def _build_batch(self, comps, capacity):
        data, weights, kernel = self._observation()
        batch = BlendBatch(data[None], weights[None], [self._specs(comps)], kernel=kernel,
                           max_iter=max(capacity, 1), device=self.device)
        if getattr(self, "_psf", None) is not None:
            # ConvolutionRenderer(psf_shift=...): the kernel moves with a free shift that
            # the device steps along with the components (smi_batch_set_kernel_shift)
            shift, renderer = self._psf
            image = np.asarray(renderer.diff_kernel.image, dtype=np.float32)
            try:
                batch.set_kernel_shift(
                    image[:kernel.shape[0]], np.asarray(shift), step=self._psf_step,
                    fft_shape=fft._get_fft_shape(image, image, padding=10, axes=(-2, -1)),
                    m=shift.m, v=shift.v, vhat=shift.vhat)
                if self._psf_rel:
                    batch.set_kernel_shift_relative_step(self._psf_rel)
            except _lib.ScarletAmdError as err:
                batch.close()
                if "fused" in str(err):
                    raise _HostSteppedShift()
                raise
        for obs, idx in self._lowres:
            _, handle, _ = obs.renderer._resampler()
            batch.attach_lowres(handle, idx, obs.data, obs.weights, obs.log_norm)
        for extra_data, extra_weights, extra_kernel in self._extra_layers:
            batch.add_observation(extra_data[None], extra_weights[None], extra_kernel)
        if self._loss_constant:
            batch.add_loss_constant(self._loss_constant)
        self._upload_state(batch, comps)
        return batch

    @staticmethod
    def _upload_state(batch, comps):
        """Warm start: the AMSGrad moments stored on the Parameters (blend.py:153-163)."""
        params = [(c.children[0]._parameters[0], c.children[1]._parameters[0]) for c in comps]
        point = [isinstance(c.children[1], PointSourceMorphology) for c in comps]

        def state(p, name, shape):
            value = getattr(p, name)
            return np.zeros(shape) if value is None else value

        if any(getattr(p, name) is not None for pair in params for p in pair
               for name in ("m", "v", "vhat")):
            # missing moments (fresh parameters) are zeros (blend.py:154-160); a point
            # source has no image on the device side: zeros of its box shape
            def image_state(name):
                return [np.zeros(batch._shapes[k]) if point[k] else state(i, name, i.shape)
                        for k, (_, i) in enumerate(params)]

            batch.set_moments(
                m_sed=np.stack([state(s_, "m", s_.shape) for s_, _ in params]),
                v_sed=np.stack([state(s_, "v", s_.shape) for s_, _ in params]),
                vhat_sed=np.stack([state(s_, "vhat", s_.shape) for s_, _ in params]),
                m_morph=image_state("m"), v_morph=image_state("v"), vhat_morph=image_state("vhat"),
            )
        vec = [None] * len(comps)  # the free 2-vector of a component, if it has one
        for k, c in enumerate(comps):
            if point[k]:
                vec[k] = params[k][1]
            elif batch.has_shift(k):
                vec[k] = c.children[1]._parameters[1]
        if any(p is not None and p.m is not None and p.v is not None and p.vhat is not None
               for p in vec):
            batch.set_center_moments(
                *[[getattr(p, name) if p is not None and getattr(p, name) is not None
                   else (0.0, 0.0) for p in vec] for name in ("m", "v", "vhat")])

    def _download(self, batch, comps):
        """Device -> the Parameters (values in place, moments as float64 arrays);
        host-updated parameters keep the host's moments."""
        self._download_all(batch, comps)
        for _, hp in getattr(self, "_host", ()):
            hp.store()
        if getattr(self, "_psf", None) is not None:
            state = batch.kernel_shift()
            shift = self._psf[0]
            shift[...] = state["shift"][0]
            shift.m, shift.v, shift.vhat = (state[n][0].copy() for n in ("m", "v", "vhat"))

    @staticmethod
    def _download_all(batch, comps):
        seds, morphs = batch.parameters()
        # (moments as float64 arrays, converted download by download: the Parameters of a
        # batch hold views)
        mom = batch.moments(dtype=np.float64)
        m_sed, v_sed, vhat_sed = mom["m_sed"], mom["v_sed"], mom["vhat_sed"]
        m_morph, v_morph, vhat_morph = mom["m_morph"], mom["v_morph"], mom["vhat_morph"]
        centers = None
        shifted = (np.asarray(batch._flags) & _lib.COMPONENT_SHIFTING).astype(bool).tolist()
        if not any(shifted) and not any(isinstance(c._children[1], PointSourceMorphology) for c in comps):
            # images only (a thousand blends: ten thousand components): nothing but assignments
            for k, comp in enumerate(comps):
                spectrum, morphology = comp._children
                sed = spectrum._parameters[0]
                image = morphology._parameters[0]
                sed[...] = seds[k]
                sed.m, sed.v, sed.vhat = m_sed[k], v_sed[k], vhat_sed[k]
                image[...] = morphs[k]
                image.m, image.v, image.vhat = m_morph[k], v_morph[k], vhat_morph[k]
            return
        for k, comp in enumerate(comps):
            spectrum, morphology = comp._children
            sed = spectrum._parameters[0]
            image = morphology._parameters[0]
            sed[...] = seds[k]
            sed.m, sed.v, sed.vhat = m_sed[k], v_sed[k], vhat_sed[k]
            if isinstance(morphology, PointSourceMorphology):
                if centers is None:
                    centers = batch.centers()
                image[...] = centers["center"][k]
                if batch.has_shift(k):  # on an ImagePSF: the device holds the offset (_point_spec)
                    bbox = morphology.bbox
                    image += np.array(bbox.origin[-2:]) + np.array(bbox.shape[-2:]) / 2
                image.m, image.v, image.vhat = (centers[n][k].copy() for n in ("m", "v", "vhat"))
                continue
            if batch.has_shift(k):
                if centers is None:
                    centers = batch.centers()
                shift = comp.children[1]._parameters[1]
                shift[...] = centers["center"][k]
                shift.m, shift.v, shift.vhat = (centers[n][k].copy() for n in ("m", "v", "vhat"))
            image[...] = morphs[k]
            image.m, image.v, image.vhat = m_morph[k], v_morph[k], vhat_morph[k]

    # --------------------------------------------------------------------- fit
    def fit(self, max_iter=200, e_rel=1e-3, min_iter=1, noise_factor=0, **alg_kwargs):
        """Fit all sources to the observation.

        Returns ``(number of loss evaluations, final logL)`` like the reference
        (blend.py:194).  Iteration order: gradient at the current parameters
        (loss appended), AMSGrad + proximal update of every parameter, every 10
        iterations the resize hook, then the convergence test
        ``it > min_iter and |dL| < e_rel |L|``.

        ``callback(*parameters, it=it)`` (an ``alg_kwargs`` entry like in the
        reference, blend.py:168,301-302) switches to host-stepped mode: one
        iteration per device call, parameters downloaded before every call of the
        callback; ``StopIteration`` raised by it ends the fit cleanly."""
        if noise_factor and (len(self.observations) != 1 or
                             tuple(self.observations[0].shape) != tuple(self.frame.shape) or
                             type(self.observations[0].renderer) is ResolutionRenderer):
            raise NotImplementedError("noise_factor > 0 needs one observation on the model frame")
        scheme = alg_kwargs.pop("scheme", "amsgrad")
        callback = alg_kwargs.pop("callback", None)
        # the device loop is AMSGrad (the reference's default); any other scheme of
        # proxmin.adaprox steps every parameter on the host from the device's gradients
        self._scheme = (scheme, alg_kwargs.pop("p", 0.25))
        prox_max_iter, opt = _adaprox_options(alg_kwargs)
        if any(type(obs.renderer) not in (NullRenderer, ConvolutionRenderer, ResolutionRenderer)
               for obs in self.observations):
            # plug-in seam (SURVEY 8b seam 3): user-written Renderer subclasses are host code
            if noise_factor or callback is not None or scheme != "amsgrad":
                raise NotImplementedError(
                    "a user-defined renderer together with noise_factor, callback or another scheme")
            return self._fit_with_host_renderers(max_iter, e_rel, min_iter, prox_max_iter, opt)
        free = [p for obs in self.observations for p in obs.parameters if not p.fixed]
        if free and scheme != "amsgrad":
            raise NotImplementedError("a free psf_shift with scheme={!r}".format(scheme))
        self._psf = None
        if free:
            self._specs(_flatten(self.sources))
            if self._host:
                raise NotImplementedError(
                    "user-defined constraints / steps together with a free psf_shift")
            if sum(any(not p.fixed for p in o.parameters) for o in self.observations) > 1:
                # one set of kernels moves on the device: the free shifts of SEVERAL
                # observations (blend.py:103-105 collects every observation's parameters)
                # are stepped on the host
                if noise_factor:
                    raise NotImplementedError("noise_factor > 0 with free psf_shifts of several observations")
                return self._fit_with_psf_shifts(max_iter, e_rel, min_iter, prox_max_iter, opt,
                                                 callback)
            self._psf = self._free_psf_shift()
            if self._psf_host is not None:
                if noise_factor:
                    raise NotImplementedError(
                        "noise_factor > 0 with a host-stepped psf_shift (prior, constraint or "
                        "step callable on it)")
                try:
                    return self._fit_with_psf_shift(max_iter, e_rel, min_iter, prox_max_iter, opt,
                                                    callback)
                finally:
                    self._psf, self._psf_stepped_on_device = None, False
        extra = () if self._psf is None else (self._psf[0],)

        from .fitting import _device_hook_covers, _fit_blends_on  # (fitting imports this module)

        if (not free and scheme == "amsgrad" and callback is None and not noise_factor
                and os.environ.get("SCARLET_AMD_BLEND_FIT") != "loop"  # development aid: A/B runs
                and all(_device_hook_covers(src) for src in self.sources)):
            # Factorized image components under the stock hooks: the batch that stays on the
            # device for the whole fit (fit_blends' path, for one blend) -- resize test and
            # resize on the device, no rebuilt batch after an UpdateException.  Quickstart
            # blend with resizing on: 34 -> 24 ms for Blend.fit(100, 1e-4), the same bits.
            out, errors = _fit_blends_on([self], self.device, max_iter, e_rel, min_iter,
                                         _from_fit=True, prox_max_iter=prox_max_iter, **opt)
            if out is not None:
                self._psf_stepped_on_device = False
                if errors:
                    raise errors[0][1]
                logger.info("scarlet ran for {0} iterations to logL = {1}".format(*out[0]))
                return out[0]

        it = 0
        while it < max_iter:
            comps = _flatten(self.sources)
            try:
                batch = self._build_batch(comps, max_iter - it)
            except _HostSteppedShift:
                # frames beyond the fused convolution kernel: the shift is stepped by the host
                if noise_factor:
                    raise NotImplementedError(
                        "noise_factor > 0 with a free psf_shift on a frame beyond the fused "
                        "convolution (the host-stepped shift does not redraw the noise)")
                try:
                    return self._fit_with_psf_shift(max_iter, e_rel, min_iter, prox_max_iter, opt,
                                                    callback)
                finally:
                    self._psf, self._psf_stepped_on_device = None, False
            batch.set_optimizer(**opt)
            restart = False
            try:
                local = 0  # adaprox's own counter, restarts after every resize
                while it + local < max_iter and not restart:
                    # the resize hook fires after the update of local iterations 10, 20, ...
                    # i.e. once 11, 21, ... iterations of this batch are done
                    next_hook = 11 if local == 0 else ((local - 1) // 10 + 1) * 10 + 1
                    n = min(next_hook - local, max_iter - it - local)
                    if callback is not None or self._host or noise_factor:
                        n = 1
                    if noise_factor:
                        self._draw_noise(batch, noise_factor)
                    # plug-in seam: gradients at the parameters of this iteration for the
                    # parameters the host updates (hoststep.py)
                    grads = self._host_gradients(batch, comps) if self._host else None
                    batch.step(local, n, e_rel=e_rel, min_iter=min_iter,
                               prox_max_iter=prox_max_iter, check_convergence=True)
                    active, err = batch.status()
                    done = len(batch.loss_history()[0])
                    if self._host and err < 0 and done == local + 1:
                        self._host_update(batch, local, grads, e_rel, prox_max_iter, opt)
                        if not all(hp.p.is_finite for _, hp in self._host):
                            err = 0
                    if err >= 0:
                        self.loss.extend(batch.loss_history()[0])
                        self._download(batch, comps)
                        raise ArithmeticError("parameters of the blend are not finite")
                    hook = done == local + n and done > 1 and (done - 1) % 10 == 0
                    local = done
                    if hook:
                        self._download(batch, comps)
                        for src in self.sources:
                            try:
                                src.update()
                            except UpdateException:
                                restart = True
                    if active == 0 and not restart:
                        break
                    if callback is not None and not restart:
                        if not hook:
                            self._download(batch, comps)
                        try:
                            callback(*self.parameters, *extra, it=local - 1)
                        except StopIteration:
                            break
                self.loss.extend(batch.loss_history()[0])
                if not restart:
                    self._download(batch, comps)
            finally:
                batch.close()
            if not restart:
                break
            it = len(self.loss)

        logger.info("scarlet ran for {0} iterations to logL = {1}".format(
            len(self.loss), -self.loss[-1]))
        for p in self.parameters + extra:
            if p.v is not None:
                p.std = STD_FROM_V  # rough estimate, blend.py:189-192
        # what _specs / _observation read is per call: nothing of this fit's renderer
        # parameters or scheme may steer a later fit_blends
        self._psf_stepped_on_device = self._psf is not None
        self._psf = None
        self._scheme = ("amsgrad", 0.25)
        return len(self.loss), -self.loss[-1]

    def _draw_noise(self, batch, noise_factor):
        """``noise_factor > 0`` (observation.py:165-168): every evaluation of the likelihood
        sees the data plus a fresh noise draw and weights / (noise_factor + 1); the
        normalisation stays that of the original weights."""
        obs = self.observations[0]
        data, weights = obs.data, obs.weights
        try:
            obs.data, obs.weights = obs.noisy(noise_factor)
            noisy_data, noisy_weights, _ = self._observation()
        finally:
            obs.data, obs.weights = data, weights
        batch.set_observation(noisy_data[None], noisy_weights[None])
        # the device derives log_norm from the weights it holds: take the scaling out again
        n_seen = int(np.count_nonzero(noisy_weights))
        batch.add_loss_constant(-0.5 * n_seen * np.log(noise_factor + 1.0))

    @staticmethod
    def _psf_shift_of(obs):
        """(renderer, shift) of an observation whose only parameter is its renderer's psf_shift"""
        renderer = obs.renderer
        shift = renderer.get_parameter("psf_shift")
        if [id(p) for p in obs.parameters] != [id(shift)] or type(renderer) is not ConvolutionRenderer:
            raise NotImplementedError("only ConvolutionRenderer(psf_shift=...) has free parameters")
        return renderer, shift

    def _free_psf_shift(self):
        """``(shift parameter, renderer)`` of the one observation whose
        ``ConvolutionRenderer(psf_shift=...)`` carries the free sub-pixel shift of the
        difference kernel (renderer.py:175-177, 215-228)."""
        free = [o for o in self.observations if any(not p.fixed for p in o.parameters)]
        if len(free) != 1:
            # (one set of kernels moves on the device: the first observation's)
            raise NotImplementedError("free psf_shifts of several observations")
        obs = free[0]
        renderer, shift = self._psf_shift_of(obs)
        if tuple(obs.shape) != tuple(self.frame.shape) or list(obs.channels) != list(self.frame.channels):
            raise NotImplementedError("psf_shift needs an observation on the model frame")
        # (relative_step, parameter.py:126-129: max(minimum, factor * mean(shift)); a prior, a
        # constraint or a step callable on the shift: the host steps it, _fit_with_psf_shift)
        on_device, rule = _vector_rule(shift, "psf_shift")
        self._psf_host = None if on_device else HostVector(shift, rule)
        const, self._psf_rel, low = rule if on_device else (0.0, 0.0, 0.0)
        self._psf_step = max(const, float(np.max(low)))
        return shift, renderer

    def _fit_with_psf_shift(self, max_iter, e_rel, min_iter, prox_max_iter, opt, callback):
        """``ConvolutionRenderer(psf_shift=...)``: the difference kernel carries a free
        sub-pixel shift (renderer.py:175-177, 215-228), for frames beyond the fused convolution
        kernel (the others step the shift on the device, ``smi_batch_set_kernel_shift``).
        Host-stepped: per iteration the
        device runs the usual step with the kernel at the current shift, and two extra
        forward renders with the kernel's derivatives give
        ``d(-logL)/d(shift) = sum w (m - d) (model (*) dK/ds)``; the shift then takes its
        unconstrained AMSGrad step (step 1e-2) on the host."""
        shift, renderer = self._free_psf_shift()
        if len(self.observations) != 1:
            raise NotImplementedError(
                "a free psf_shift with several observations on a frame beyond the fused convolution")
        obs = self.observations[0]
        alpha0, rel = self._psf_step, self._psf_rel
        self._psf = None  # _specs / _download: no device-side shift in this mode
        C = self.frame.C
        data = np.ascontiguousarray(obs.data, dtype=np.float32)
        weights = np.ascontiguousarray(obs.weights, dtype=np.float32)

        def stamp(k):  # (Ck, p, q) -> (C, odd p, odd q), centres aligned
            k = np.asarray(k, dtype=np.float32)
            ph, pw = k.shape[1] | 1, k.shape[2] | 1
            out = np.zeros((C, ph, pw), dtype=np.float32)
            oy, ox = ph // 2 - k.shape[1] // 2, pw // 2 - k.shape[2] // 2
            out[:, oy:oy + k.shape[1], ox:ox + k.shape[2]] = k
            return out

        # (the step itself: hoststep.HostVector -- amsgrad_pair of the device in float64, plus
        # the shift's prior / constraint / step callable if it has any)
        stepper = getattr(self, "_psf_host", None) or HostVector(shift, (alpha0, rel, 0.0))
        it = 0
        while it < max_iter:
            comps = _flatten(self.sources)
            batch = BlendBatch(data[None], weights[None], [self._specs(comps)],
                               kernel=stamp(renderer.kernel_image()), max_iter=max(max_iter - it, 1),
                               device=self.device)
            self._upload_state(batch, comps)
            batch.set_optimizer(**opt)
            restart = False
            try:
                local = 0
                while it + local < max_iter and not restart:
                    # gradient w.r.t. the shift at the parameters of this iteration
                    _, rendered, _ = batch.forward(model=False)
                    resid = weights.astype(np.float64) * (rendered[0] - data)
                    g = np.zeros(2)
                    for a, dk in enumerate(renderer.kernel_derivatives()):
                        batch.set_kernel(stamp(dk))
                        g[a] = np.sum(resid * batch.forward(model=False)[1][0])
                    batch.set_kernel(stamp(renderer.kernel_image()))
                    batch.step(local, 1, e_rel=e_rel, min_iter=min_iter,
                               prox_max_iter=prox_max_iter, check_convergence=True)
                    # AMSGrad (lite/parameters.py:274-291), sub-iterations if it is constrained
                    stepper.update(local, g, e_rel, prox_max_iter, **opt)
                    stepper.store()
                    batch.set_kernel(stamp(renderer.kernel_image()))
                    active, err = batch.status()
                    done = len(batch.loss_history()[0])
                    if err >= 0 or not np.all(np.isfinite(np.asarray(shift))):
                        self.loss.extend(batch.loss_history()[0])
                        self._download(batch, comps)
                        raise ArithmeticError("parameters of the blend are not finite")
                    hook = done == local + 1 and done > 1 and (done - 1) % 10 == 0
                    local = done
                    if hook:
                        self._download(batch, comps)
                        for src in self.sources:
                            try:
                                src.update()
                            except UpdateException:
                                restart = True
                    if active == 0 and not restart:
                        break
                    if callback is not None and not restart:
                        if not hook:
                            self._download(batch, comps)
                        try:
                            callback(*self.parameters, shift, it=local - 1)
                        except StopIteration:
                            break
                self.loss.extend(batch.loss_history()[0])
                if not restart:
                    self._download(batch, comps)
            finally:
                batch.close()
            if not restart:
                break
            it = len(self.loss)
        for p in self.parameters + (shift,):
            if p.v is not None:
                p.std = STD_FROM_V
        return len(self.loss), -self.loss[-1]

    def _fit_with_psf_shifts(self, max_iter, e_rel, min_iter, prox_max_iter, opt, callback):
        """Free ``psf_shift``s of SEVERAL observations (blend.py:103-105: the parameters of
        every observation join the sources' in ``X``).  The device moves one set of kernels,
        so here every shift is the host's: the observations -- on the model's grid, each with
        channels of its own -- are merged into one cube as usual (``_observation``), per
        iteration the device runs its step with every kernel at its current shift, and for
        each free shift two more forward renders with that observation's ``dK/ds`` in its
        channels (zero kernels elsewhere) give ``d(-logL)/ds = sum w (m - d) (model (*) dK/ds)``
        over its channels; the shifts take their AMSGrad steps (``hoststep.HostVector``: also
        with a prior, a constraint or a step callable) and the merged kernel cube is rebuilt."""
        channels = list(self.frame.channels)
        C = self.frame.C
        movers = []
        for obs in self.observations:
            free = [p for p in obs.parameters if not p.fixed]
            if not free:
                continue
            renderer, shift = self._psf_shift_of(obs)
            _, rule = _vector_rule(shift, "psf_shift")
            movers.append((obs, renderer, shift, [channels.index(c) for c in obs.channels],
                           HostVector(shift, rule)))

        def merged():
            """(data, weights, kernel cube with C bands) at the current shifts"""
            self._psf = True  # (lets _observation accept free renderer parameters)
            try:
                data, weights, kernel = self._observation()
            finally:
                self._psf = None
            if self._lowres or self._extra_layers or kernel is None:
                raise NotImplementedError(
                    "free psf_shifts of several observations need observations on the model's "
                    "grid with channels of their own")
            kernel = np.asarray(kernel, dtype=np.float32)
            return data, weights, (np.repeat(kernel, C, axis=0) if kernel.shape[0] == 1 else kernel)

        data, weights, kernel = merged()
        ph, pw = kernel.shape[-2:]

        def derivative_cube(idx, dk):
            """dK/ds of one observation in its channels, zero kernels in the others"""
            dk = np.asarray(dk, dtype=np.float32)
            cube = np.zeros((C, ph, pw), dtype=np.float32)
            oy, ox = ph // 2 - dk.shape[1] // 2, pw // 2 - dk.shape[2] // 2
            for j, c in enumerate(idx):
                cube[c, oy:oy + dk.shape[1], ox:ox + dk.shape[2]] = dk[j if dk.shape[0] > 1 else 0]
            return cube

        shifts = tuple(m[2] for m in movers)
        w64 = weights.astype(np.float64)
        it = 0
        while it < max_iter:
            comps = _flatten(self.sources)
            batch = BlendBatch(data[None], weights[None], [self._specs(comps)], kernel=kernel,
                               max_iter=max(max_iter - it, 1), device=self.device)
            self._upload_state(batch, comps)
            batch.set_optimizer(**opt)
            restart = False
            try:
                local = 0
                while it + local < max_iter and not restart:
                    _, rendered, _ = batch.forward(model=False)
                    resid = w64 * (rendered[0] - data)
                    grads = []
                    for obs, renderer, shift, idx, stepper in movers:
                        g = np.zeros(2)
                        for a, dk in enumerate(renderer.kernel_derivatives()):
                            batch.set_kernel(derivative_cube(idx, dk))
                            g[a] = np.sum(resid[idx] * batch.forward(model=False)[1][0][idx])
                        grads.append(g)
                    batch.set_kernel(kernel)
                    batch.step(local, 1, e_rel=e_rel, min_iter=min_iter,
                               prox_max_iter=prox_max_iter, check_convergence=True)
                    for (obs, renderer, shift, idx, stepper), g in zip(movers, grads):
                        stepper.update(local, g, e_rel, prox_max_iter, **opt)
                        stepper.store()
                    kernel = merged()[2]
                    batch.set_kernel(kernel)
                    active, err = batch.status()
                    done = len(batch.loss_history()[0])
                    if err >= 0 or not all(np.all(np.isfinite(np.asarray(s))) for s in shifts):
                        self.loss.extend(batch.loss_history()[0])
                        self._download(batch, comps)
                        raise ArithmeticError("parameters of the blend are not finite")
                    hook = done == local + 1 and done > 1 and (done - 1) % 10 == 0
                    local = done
                    if hook:
                        self._download(batch, comps)
                        for src in self.sources:
                            try:
                                src.update()
                            except UpdateException:
                                restart = True
                    if active == 0 and not restart:
                        break
                    if callback is not None and not restart:
                        if not hook:
                            self._download(batch, comps)
                        try:
                            callback(*self.parameters, *shifts, it=local - 1)
                        except StopIteration:
                            break
                self.loss.extend(batch.loss_history()[0])
                if not restart:
                    self._download(batch, comps)
            finally:
                batch.close()
            if not restart:
                break
            it = len(self.loss)
        for p in self.parameters + shifts:
            if p.v is not None:
                p.std = STD_FROM_V
        self._psf_stepped_on_device = False
        return len(self.loss), -self.loss[-1]

    def _host_render_ops(self):
        """Per observation ``(obs, forward, adjoint, log_norm)`` for the host-rendered mode:
        ``forward(model cube) -> observation frame`` and its transpose, both float64.
        Built-in renderers have theirs; a user renderer supplies ``adjoint``.  Every pair is
        checked with a random dot product: a renderer that is not linear, or whose ``adjoint``
        is not its transpose, is refused."""
        ops = []
        rng = np.random.default_rng(0)
        for obs in self.observations:
            r = obs.renderer
            if any(not p.fixed for p in obs.parameters):
                raise NotImplementedError("free renderer parameters in the host-rendered mode")
            if type(r) is ResolutionRenderer:
                raise NotImplementedError(
                    "a ResolutionRenderer observation next to a user-defined renderer")
            if type(r) in (NullRenderer, ConvolutionRenderer):
                data_sl, model_sl = r.slices
                kernel = None if type(r) is NullRenderer else np.asarray(r.kernel_image(), np.float64)
                if kernel is not None and (kernel.shape[1] % 2 == 0 or kernel.shape[2] % 2 == 0):
                    # the transpose below flips the stamp about its centre pixel: an even side
                    # gets a zero row / column behind it (the centre h // 2 stays the centre),
                    # like _observation does for the device
                    kernel = np.pad(kernel, ((0, 0), (0, 1 - kernel.shape[1] % 2),
                                             (0, 1 - kernel.shape[2] % 2)))

                def forward(model, r=r, kernel=kernel, data_sl=data_sl, model_sl=model_sl):
                    m = np.asarray(r.map_channels(model), dtype=np.float64)
                    if kernel is not None:
                        m = fft.convolve(fft.Fourier(m), kernel, axes=(1, 2)).image
                    out = np.zeros(r.data_frame.shape, dtype=np.float64)
                    out[data_sl] = m[model_sl]
                    return out

                def adjoint(res, r=r, kernel=kernel, data_sl=data_sl, model_sl=model_sl):
                    idx = [list(self.frame.channels).index(c) for c in r.data_frame.channels]
                    back = np.zeros((len(idx),) + tuple(self.frame.shape[1:]), dtype=np.float64)
                    back[model_sl] = np.asarray(res, dtype=np.float64)[data_sl]
                    if kernel is not None:  # transpose of a convolution: the flipped kernel
                        back = fft.convolve(fft.Fourier(back), kernel[:, ::-1, ::-1],
                                                 axes=(1, 2)).image
                    g = np.zeros(self.frame.shape, dtype=np.float64)
                    g[idx] = back
                    return g
            else:
                if not callable(getattr(r, "adjoint", None)):
                    raise NotImplementedError(
                        "renderer {}: the reference differentiates a user-defined renderer "
                        "automatically; here it must be linear and provide "
                        "`adjoint(residual) -> gradient image in the model frame`".format(
                            type(r).__name__))

                def forward(model, obs=obs):
                    return np.asarray(obs.render(model), dtype=np.float64)

                def adjoint(res, r=r):
                    return np.asarray(r.adjoint(np.asarray(res, dtype=np.float64)), dtype=np.float64)

            # <R x, y> = <x, R^T y> and R(2x) = 2 R(x), for the built-in pairs above as well
            x = rng.standard_normal(self.frame.shape)
            y = rng.standard_normal(obs.data.shape)
            fx = forward(x)
            lhs, rhs = float(np.sum(fx * y)), float(np.sum(x * adjoint(y)))
            lin = forward(2.0 * x) - 2.0 * fx
            if (abs(lhs - rhs) > 1e-6 * (abs(lhs) + abs(rhs)) + 1e-12
                    or np.abs(lin).max() > 1e-6 * np.abs(fx).max() + 1e-12):
                raise NotImplementedError(
                    "renderer {} is not linear, or its `adjoint` is not the transpose of its "
                    "forward map (<R x, y> = {:.6g}, <x, R^T y> = {:.6g})".format(
                        type(r).__name__, lhs, rhs))
            ops.append((obs, forward, adjoint, float(obs.log_norm)))
        return ops

    def _fit_with_host_renderers(self, max_iter, e_rel, min_iter, prox_max_iter, opt):
        """The fit with user-written ``Renderer`` subclasses (observation.py:59-112 accepts
        any; renderer.py:12-24).  They are host code, so per iteration the device renders
        the model cube, the host maps it into every observation, forms the loss
        ``sum_obs log_norm + 1/2 sum w (R m - d)^2`` (blend.py:259-274) and pulls the weighted
        residuals back through the transposes, ``g = sum_obs R^T w (R m - d)``; the device
        then takes the usual step -- gradient gather, AMSGrad, proximal sub-iterations --
        with that gradient image: its identity-renderer likelihood is handed the stand-in
        observation ``d' = m - g / w'`` with constant weight ``w' = 2^-20``, whose residual
        ``w' (m - d')`` is ``g`` (the scale keeps ``d'`` dominated by ``g``, so the subtraction
        loses nothing).  Loss history, stopping rule (blend.py:294-299) and the resize hook
        are the host's in this mode."""
        ops = self._host_render_ops()
        self._psf = None
        scale = np.float32(2.0 ** -20)
        ones = np.full((1,) + tuple(self.frame.shape), scale, dtype=np.float32)
        it = 0
        stop = False
        while it < max_iter and not stop:
            comps = _flatten(self.sources)
            specs = self._specs(comps)
            if self._host:
                raise NotImplementedError(
                    "user-defined constraints / steps together with a user-defined renderer")
            batch = BlendBatch(np.zeros_like(ones), ones, [specs], kernel=None,
                               max_iter=max(max_iter - it, 1), device=self.device, log_norm=False)
            self._upload_state(batch, comps)
            batch.set_optimizer(**opt)
            restart = False
            try:
                local = 0
                while it + local < max_iter and not restart:
                    model = batch.forward(rendered=False)[0][0].astype(np.float64)
                    loss, g = 0.0, np.zeros(self.frame.shape, dtype=np.float64)
                    for obs, forward, adjoint, log_norm in ops:
                        res = forward(model) - obs.data
                        wres = obs.weights * res
                        loss += log_norm + 0.5 * float(np.sum(wres * res))
                        g += adjoint(wres)
                    self.loss.append(loss)
                    batch.set_observation((model - g / float(scale)).astype(np.float32)[None], ones)
                    batch.step(local, 1, e_rel=e_rel, min_iter=min_iter,
                               prox_max_iter=prox_max_iter, check_convergence=False)
                    active, err = batch.status()
                    if err >= 0 or not np.isfinite(loss):
                        self._download(batch, comps)
                        raise ArithmeticError("parameters of the blend are not finite")
                    local += 1
                    if local > 1 and (local - 1) % 10 == 0:  # blend.py:284-292
                        self._download(batch, comps)
                        for src in self.sources:
                            try:
                                src.update()
                            except UpdateException:
                                restart = True
                    n = len(self.loss)
                    if (not restart and local - 1 > min_iter and n > 1
                            and abs(self.loss[-2] - self.loss[-1]) < e_rel * abs(self.loss[-1])):
                        stop = True
                        break
                if not restart:
                    self._download(batch, comps)
            finally:
                batch.close()
            if not restart:
                break
            it = len(self.loss)
        for p in self.parameters:
            if p.v is not None:
                p.std = STD_FROM_V
        self._scheme = ("amsgrad", 0.25)
        return len(self.loss), -self.loss[-1]

    # ------------------------------------------------------------------- model
    def get_model(self, *parameters, frame=None):
        """(C, H, W) model cube: every source's boxed model added into a zero cube
        of the frame's dtype (host evaluation for inspection; blend.py:200-244)."""
        models = self.get_models_of_children(*parameters, frame=None)
        if frame is None:
            frame = self.frame
        full = np.zeros(frame.shape, dtype=frame.dtype)
        for src, model in zip(self.sources, models):
            if frame == self.frame:
                fs, ms = src._model_frame_slices, src._model_slices
            else:
                fs, ms = overlapped_slices(frame.bbox, src.bbox)
            full[fs] += model[ms]
        return full

    @property
    def log_likelihood(self):
        return -np.array(self.loss)

    @property
    def bbox(self):
        return self.frame.bbox


def __getattr__(name):
    """``fit_blends`` and its helpers live in ``scarlet_amd.fitting`` (which imports this module);
    the names they had here keep working: ``from scarlet_amd.blend import fit_blends``."""
    if name in ("fit_blends", "_fit_blends_on", "_fit_group_resident", "_fit_group_rebuilt",
                "_device_resize_covers", "_device_hook_covers", "_resized_spec", "_refresh_boxes",
                "_export_state", "_import_state", "_standard_size"):
        from . import fitting

        return getattr(fitting, name)
    raise AttributeError("module {!r} has no attribute {!r}".format(__name__, name))
