"""Tree of parameterised models (reference scarlet/model.py): every node owns
``Parameter``s and child models; ``parameters`` flattens the tree depth first,
own parameters before the children's -- the order ``Blend.fit`` relies on."""

from abc import ABC, abstractmethod

from .parameter import Parameter


class UpdateException(Exception):
    """Raised by ``update()`` when the optimizer has to be restarted (box resize)."""


class Model(ABC):
    def __init__(self, *parameters, children=None):
        for p in parameters:
            assert isinstance(p, Parameter)
        self._parameters = tuple(parameters)
        if children is None:
            children = ()
        if not hasattr(children, "__iter__"):
            children = (children,)
        for c in children:
            assert isinstance(c, Model)
        self._children = children
        self.check_parameters()

    @property
    def parameters(self):
        own = tuple(self._parameters)
        return own + tuple(p for c in self.children for p in c.parameters)

    @property
    def children(self):
        return self._children

    def __getitem__(self, i):
        return self._children[i]

    def __iter__(self):
        return iter(self._children)

    def get_parameter(self, i, *parameters):
        """Parameter by position, slice or ``name`` among ``parameters`` (or
        among this model's own when none are passed)."""
        pool = parameters if parameters else self.parameters
        if isinstance(i, (int, slice)):
            return pool[i]
        if isinstance(i, str):
            hits = tuple(p for p in pool if isinstance(p, Parameter) and p.name == i)
            if not hits:
                return None
            return hits[0] if len(hits) == 1 else hits
        return None

    @abstractmethod
    def get_model(self, *parameters, **kwargs):
        """Realisation of the model for the given (or the stored) parameters."""

    def get_models_of_children(self, *parameters, **kwargs):
        models = []
        if parameters:
            i = len(self._parameters)
            for c in self._children:
                j = len(c.parameters)
                models.append(c.get_model(*parameters[i : i + j], **kwargs))
                i += j
        else:
            models = [c.get_model(**kwargs) for c in self._children]
        return models

    def check_parameters(self):
        """``ArithmeticError`` if any parameter holds a non-finite value."""
        for p in self.parameters:
            if not p.is_finite:
                raise ArithmeticError(
                    "Model {}, Parameter '{}' is not finite:\n{}".format(
                        type(self).__name__, p.name, p
                    )
                )

    def update(self):
        """Hook for state changes outside the gradient path (e.g. box resizing)."""
