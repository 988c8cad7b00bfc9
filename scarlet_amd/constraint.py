"""Constraints = proximal operators ``f(X, step) -> X'`` with the reference's
class names and defaults (scarlet/constraint.py).

Called stand-alone they act on host arrays (the monotonicity sweep goes through
the C ABI to the GPU).  Inside ``Blend.fit`` the built-in chains are not called
at all: the chain of a parameter is translated into ``SMI_PROX_*`` flags and runs
fused in the device update kernel (``device_flags`` below).  A chain the device
cannot express -- a user subclass, another order -- is called
as written, on the host, for that parameter only (``hoststep.py``).
"""

import numpy as np

from . import _lib, operator
from .cache import Cache


class Constraint:
    """Base class; wraps any proximal mapping ``f(X, step)``."""

    def __init__(self, f=None):
        self.f = f

    def __call__(self, X, step):
        return X if self.f is None else self.f(X, step)


class ConstraintChain:
    """Constraints applied one after another, ``repeat`` times."""

    def __init__(self, *constraints, repeat=1):
        assert isinstance(repeat, int) and repeat >= 1
        self.constraints = constraints
        self.repeat = repeat

    def __call__(self, X, step):
        for _ in range(self.repeat):
            for c in self.constraints:
                X = c(X, step)
        return X


class PositivityConstraint(Constraint):
    """Values not smaller than ``zero``."""

    def __init__(self, zero=0):
        self.zero = zero

    def __call__(self, X, step):
        return np.maximum(X, self.zero)


class NormalizationConstraint(Constraint):
    """Scale ``X`` in place so that its sum or its maximum is one."""

    def __init__(self, type="sum"):
        type = type.lower()
        assert type in ["sum", "max"]
        self.type = type

    def __call__(self, X, step):
        X /= X.sum() if self.type == "sum" else X.max()
        return X


def _threshold(step, thresh, type):
    assert type in ["relative", "absolute"]
    return thresh * step if type == "relative" else thresh


class L0Constraint(Constraint):
    """Hard thresholding (the reference delegates to ``proxmin.prox_hard``):
    entries with ``|x| < t`` are set to zero, in place; ``t = thresh * step``
    for ``type="relative"``."""

    def __init__(self, thresh, type="absolute"):
        self.thresh, self.type = thresh, type
        super().__init__(self._prox)

    def _prox(self, X, step):
        X[np.abs(X) < _threshold(step, self.thresh, self.type)] = 0
        return X


class L1Constraint(Constraint):
    """Soft thresholding ``sign(x) max(|x| - t, 0)`` (``proxmin.prox_soft``)."""

    def __init__(self, thresh, type="absolute"):
        self.thresh, self.type = thresh, type
        super().__init__(self._prox)

    def _prox(self, X, step):
        t = _threshold(step, self.thresh, self.type)
        return np.sign(X) * np.maximum(np.abs(X) - t, 0)


class ThresholdConstraint(Constraint):
    """Cut pixels below the last gap in the log-histogram of the positive pixel
    values (reference constraint.py:148-180), then ``prox_hard_plus``."""

    def __call__(self, X, step):
        thresh, _ = self.threshold(X)
        X[X < thresh] = 0
        return X

    def threshold(self, morph):
        positive = morph[morph > 0]
        bins = 50
        if positive.size < 500:
            bins = max(int(positive.size / 10), 1)
            if bins == 1:
                return 0, bins
        hist, edges = np.histogram(np.log10(positive).reshape(-1), bins)
        gaps = np.where(hist == 0)[0]
        if len(gaps) == 0:
            return 0, bins
        return 10 ** edges[gaps[-1]], bins


class MonotonicityConstraint(Constraint):
    """Morphology decreases monotonically away from the centre pixel
    ``(h//2, w//2)``; see ``operator.prox_weighted_monotonic``.

    Note the defaults (``"flat"``, 0.1) differ from what
    ``ExtendedSourceMorphology`` passes (``"angle"``, 0), as in the reference.
    """

    def __init__(self, neighbor_weight="flat", min_gradient=0.1, use_mask=False,
                 fit_center_radius=0):
        self.neighbor_weight = neighbor_weight
        self.min_gradient = min_gradient
        self.use_mask = use_mask
        self.fit_center = fit_center_radius > 0
        self.fit_center_radius = fit_center_radius

    def __call__(self, morph, step):
        shape = tuple(morph.shape)
        nominal = tuple(n // 2 for n in shape)
        # optionally re-centre on the brightest pixel near the nominal centre
        center = (operator.get_center(morph, nominal, radius=self.fit_center_radius)
                  if self.fit_center else nominal)
        name = "operator.prox_weighted_monotonic"
        key = (shape, center, self.neighbor_weight, self.min_gradient)
        try:
            prox = Cache.check(name, key)
        except KeyError:
            prox = operator.prox_weighted_monotonic(
                shape, neighbor_weight=self.neighbor_weight,
                min_gradient=self.min_gradient, center=center,
            )
            Cache.set(name, key, prox)
        original = morph.copy()
        result = prox(morph, step)
        if self.use_mask:
            # pixels connected to the centre by a monotonic path keep their value
            # (constraint.py:227-232)
            valid, masked, _ = operator.prox_monotonic_mask(
                original, step, center=center, center_radius=0, variance=0, max_iter=0)
            result[valid] = masked[valid]
        return result


class MonotonicMaskConstraint(Constraint):
    """Monotonicity by branching out from ``center``: pixels that no monotonic path
    connects to the centre are interpolated (``operator.prox_monotonic_mask``; reference
    constraint.py:237-259).  A 3-D ``morph`` is treated image by image.  Inside
    ``Blend.fit`` it runs on the host (hoststep.py)."""

    def __init__(self, center, center_radius=1, variance=0.0, max_iter=3):
        self.center = center
        self.center_radius = center_radius
        self.variance = variance
        self.max_iter = max_iter

    def _one(self, image, step):
        return operator.prox_monotonic_mask(
            image, step, center=self.center, center_radius=self.center_radius,
            variance=self.variance, max_iter=self.max_iter)[1]

    def __call__(self, morph, step):
        if np.ndim(morph) == 2:
            return self._one(morph, step)
        return np.array([self._one(image, step) for image in morph])


class SymmetryConstraint(Constraint):
    """Two-fold rotation symmetry about the centre, softened by ``strength``."""

    def __init__(self, strength=1):
        self.strength = strength

    def __call__(self, morph, step):
        return operator.prox_soft_symmetry(morph, step, strength=self.strength)


class CenterOnConstraint(Constraint):
    """Keep the centre pixel at least ``tiny`` so a source cannot vanish."""

    def __init__(self, tiny=1e-6):
        self.tiny = tiny

    def __call__(self, morph, step):
        c = (morph.shape[0] // 2, morph.shape[1] // 2)
        morph[c] = max(morph[c], self.tiny)
        return morph


class LeakyConstraint(Constraint):
    """``(1 - leak) * constraint(x) + leak * x``."""

    def __init__(self, constraint, leak=0.05):
        self.constraint = constraint
        self.leak = leak

    def __call__(self, x, step):
        return (1 - self.leak) * self.constraint(x, step) + self.leak * x


# canonical order of the fused device chain (morphology.py:644-670)
_DEVICE_ORDER = (
    MonotonicityConstraint, SymmetryConstraint, L0Constraint, L1Constraint,
    PositivityConstraint, CenterOnConstraint, NormalizationConstraint,
)


_DEVICE_RANK = {cls: rank for rank, cls in enumerate(_DEVICE_ORDER)}


def device_flags(constraint):
    """Translate a constraint (chain) into the fused device chain.

    Returns ``dict(flags, neighbor_weight, min_gradient, zero, l_thresh)`` or
    raises ``NotImplementedError`` for chains the device kernel cannot express
    (user callables, other orders); ``Blend`` then keeps that parameter's update on
    the host (``hoststep.HostParameter``).
    """
    out = dict(flags=0, neighbor_weight=None, min_gradient=0.0, zero=0.0, l_thresh=0.0,
               center_floor=1e-6, sym_strength=1.0, chain_repeat=1)
    if constraint is None:
        return out
    if type(constraint) is ConstraintChain or isinstance(constraint, ConstraintChain):
        out["chain_repeat"] = int(constraint.repeat)
        items = list(constraint.constraints)
    else:
        items = [constraint]
    rank = -1
    for c in items:
        r = _DEVICE_RANK.get(type(c))
        if r is None:
            raise NotImplementedError(
                "constraint {} has no device implementation".format(type(c).__name__)
            )
        if r <= rank:
            raise NotImplementedError(
                "constraint order {} is not the fused device order".format(
                    [type(i).__name__ for i in items]
                )
            )
        rank = r
        t = type(c)  # (exact types: _DEVICE_RANK has no entry for subclasses)
        if t is MonotonicityConstraint:
            if c.fit_center_radius > 1:
                raise NotImplementedError("fit_center_radius > 1 is not supported on the device")
            out["flags"] |= _lib.PROX_MONOTONIC
            if c.fit_center:
                out["flags"] |= _lib.PROX_FIT_CENTER
            if c.use_mask:
                out["flags"] |= _lib.PROX_MONO_MASK
            out["neighbor_weight"] = c.neighbor_weight
            out["min_gradient"] = float(c.min_gradient)
        elif t is SymmetryConstraint:
            out["flags"] |= _lib.PROX_SYMMETRY
            out["sym_strength"] = float(c.strength)
        elif t is L0Constraint or t is L1Constraint:
            if out["flags"] & (_lib.PROX_L0 | _lib.PROX_L1):
                raise NotImplementedError("L0 and L1 constraints in one chain")
            if c.type == "relative":
                out["flags"] |= _lib.PROX_L_RELATIVE
            out["flags"] |= _lib.PROX_L0 if t is L0Constraint else _lib.PROX_L1
            out["l_thresh"] = float(c.thresh)
        elif t is PositivityConstraint:
            out["flags"] |= _lib.PROX_POSITIVE
            out["zero"] = float(c.zero)
        elif t is CenterOnConstraint:
            out["flags"] |= _lib.PROX_CENTER_ON
            out["center_floor"] = float(c.tiny)
        elif t is NormalizationConstraint:
            out["flags"] |= _lib.PROX_NORM_MAX if c.type == "max" else _lib.PROX_NORM_SUM
    return out
