"""Tree of parameterised models.

API of the reference's ``scarlet/model.py:6-166``: every node owns ``Parameter``s and
child models; ``parameters`` flattens the tree depth first, own parameters before the
children's.  ``Blend.fit`` relies on that order to map the device state back onto the
tree, and ``get_models_of_children`` uses it to hand each child its slice of a flat
parameter list.
"""

from abc import ABC, abstractmethod
from itertools import accumulate

from .parameter import Parameter


class UpdateException(Exception):
    """Raised by ``update()`` when the optimizer has to be restarted (box resize)."""


def _as_tuple(children):
    if children is None:
        return ()
    return tuple(children) if hasattr(children, "__iter__") else (children,)


class Model(ABC):
    def __init__(self, *parameters, children=None):
        assert all(isinstance(p, Parameter) for p in parameters)
        self._parameters = tuple(parameters)
        # a list of children stays the caller's list object (sources are appended to
        # blends that way); anything else becomes a tuple
        self._children = children if isinstance(children, list) else _as_tuple(children)
        assert all(isinstance(c, Model) for c in self._children)
        self.check_parameters()

    @property
    def parameters(self):
        flat = list(self._parameters)
        for child in self._children:
            flat.extend(child.parameters)
        return tuple(flat)

    @property
    def children(self):
        return self._children

    def __getitem__(self, i):
        return self._children[i]

    def __iter__(self):
        return iter(self._children)

    def get_parameter(self, i, *parameters):
        """Parameter by position, slice or ``name`` among ``parameters`` (or among this
        model's own when none are passed).  A name yields the parameter, a tuple when
        several carry it, ``None`` when none does."""
        pool = parameters or self.parameters
        if isinstance(i, str):
            named = tuple(p for p in pool if isinstance(p, Parameter) and p.name == i)
            if len(named) > 1:
                return named
            return named[0] if named else None
        return pool[i] if isinstance(i, (int, slice)) else None

    @abstractmethod
    def get_model(self, *parameters, **kwargs):
        """Realisation of the model for the given (or the stored) parameters."""

    def get_models_of_children(self, *parameters, **kwargs):
        """Each child's model; a flat ``parameters`` list (own parameters first, then
        the children's, as ``self.parameters`` orders them) is split among them."""
        if not parameters:
            return [child.get_model(**kwargs) for child in self._children]
        counts = [len(child.parameters) for child in self._children]
        starts = accumulate([len(self._parameters)] + counts[:-1])
        return [child.get_model(*parameters[lo:lo + n], **kwargs)
                for child, lo, n in zip(self._children, starts, counts)]

    def check_parameters(self):
        """``ArithmeticError`` if any parameter holds a non-finite value."""
        broken = next((p for p in self.parameters if not p.is_finite), None)
        if broken is not None:
            raise ArithmeticError("Model {}, Parameter '{}' is not finite:\n{}".format(
                type(self).__name__, broken.name, broken))

    def update(self):
        """Hook for state changes outside the gradient path (e.g. box resizing)."""
