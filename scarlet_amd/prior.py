"""Prior base class (reference scarlet/prior.py): a prior contributes the
gradient of its negative log to the likelihood gradient.  The reference ships
no concrete priors; parameters with a prior are not supported by the device
loop and make ``Blend.fit`` raise."""

from abc import ABC, abstractmethod


class Prior(ABC):
    @abstractmethod
    def __call__(self, x):
        """Value of the negative log prior at ``x``."""

    @abstractmethod
    def grad(self, x):
        """Gradient of the negative log prior at ``x``."""
