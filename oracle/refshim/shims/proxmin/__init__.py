"""Container-only stand-in for `proxmin`: the operator definitions are taken
from oracle.proxops (our restatement); `adaprox` is deliberately absent."""
from . import operators, utils, algorithms  # noqa: F401


def adaprox(*args, **kwargs):
    raise NotImplementedError("proxmin is not available in this container")
