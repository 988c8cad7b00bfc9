"""Optimization parameters: an ``ndarray`` subclass that carries its prior,
constraint, step rule and the optimizer state (``m, v, vhat``) of the last fit,
with the reference's attribute names (scarlet/parameter.py:9-71) so that warm
starts, pickling and user code that inspects ``p.std`` keep working."""

import numpy as np
import numpy.ma as ma

from .constraint import Constraint, ConstraintChain
from .prior import Prior

_ATTRS = (
    ("name", "unnamed"), ("prior", None), ("constraint", None), ("step", 0),
    ("std", None), ("m", None), ("v", None), ("vhat", None), ("fixed", False),
)


def std_estimate(v):
    """``1 / sqrt(masked_equal(v, 0))`` (the rough error estimate of blend.py:189-192): the same
    masked array -- mask, fill value and unmasked values -- without numpy.ma's per-operation
    bookkeeping."""
    v = np.asarray(v)
    mask = v == 0
    with np.errstate(divide="ignore"):
        data = 1 / np.sqrt(v)
    return ma.array(data, mask=mask, fill_value=0.0)


# value of Parameter.std that stands for "std_estimate(self.v), when somebody asks"
STD_FROM_V = "1/sqrt(v)"


class Parameter(np.ndarray):
    """Array of parameter values plus optimization metadata.

    ``step`` is a number or a callable ``step(X, it) -> float or array``;
    ``constraint`` a ``Constraint``/``ConstraintChain`` applied as proximal
    operator; ``m, v, vhat`` the AMSGrad moments (set by ``Blend.fit``);
    ``std`` the rough error estimate ``1/sqrt(v)`` (blend.py:189-192).
    """

    def __new__(cls, array, name="unnamed", prior=None, constraint=None, step=0,
                std=None, m=None, v=None, vhat=None, fixed=False):
        obj = np.asarray(array, dtype=array.dtype).view(cls)
        if prior is not None:
            assert isinstance(prior, Prior)
        if constraint is not None:
            assert isinstance(constraint, (Constraint, ConstraintChain))
        obj.name, obj.prior, obj.constraint, obj.step = name, prior, constraint, step
        obj.std, obj.m, obj.v, obj.vhat, obj.fixed = std, m, v, vhat, fixed
        return obj

    def __array_finalize__(self, obj):
        if obj is None:
            return
        for attr, default in _ATTRS:
            if attr == "std":
                # the stored value as it is (the `STD_FROM_V` marker included): a view, a
                # slice or a comparison of a fitted Parameter must not make the estimate
                self.__dict__["_std"] = getattr(obj, "__dict__", {}).get("_std", default)
            else:
                setattr(self, attr, getattr(obj, attr, default))

    def __reduce__(self):
        base = super().__reduce__()
        return base[0], base[1], base[2] + (self.__dict__,)

    def __setstate__(self, state):
        attrs = dict(state[-1])
        if "std" in attrs:  # pickles from before `std` became a property
            attrs["_std"] = attrs.pop("std")
        self.__dict__.update(attrs)
        super().__setstate__(state[:-1])

    @property
    def std(self):
        """Rough error estimate.  A fit leaves ``STD_FROM_V`` here instead of twenty thousand
        masked arrays nobody may look at: the estimate is made from ``v`` on first access."""
        value = self.__dict__.get("_std")
        if isinstance(value, str):
            value = self.__dict__["_std"] = None if self.v is None else std_estimate(self.v)
        return value

    @std.setter
    def std(self, value):
        self.__dict__["_std"] = value

    @property
    def _data(self):
        return self.view(np.ndarray)

    @property
    def is_finite(self):
        return bool(np.isfinite(self._data).all())


def prepare_param(X, name, fixed=True, step=None):
    """Wrap scalars / sequences into a float ``Parameter`` called ``name``."""
    if isinstance(X, Parameter):
        assert X.name == name
        return X
    if np.isscalar(X):
        X = (X,)
    return Parameter(np.array(X, dtype="float"), name=name, fixed=fixed, step=step)


def relative_step(X, it, factor=0.1, minimum=0, axis=None):
    """Step = ``factor`` x mean of ``X`` along ``axis``, floored at ``minimum``."""
    return np.maximum(minimum, factor * X.mean(axis=axis))
