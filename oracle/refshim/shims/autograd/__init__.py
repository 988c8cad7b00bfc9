"""Container-only stand-in for the `autograd` package (NOT automatic
differentiation): just enough names for the reference's *forward* code to
import and run on plain NumPy.  Used only by oracle/refshim/make_golden.py in
the build container to produce golden vectors; never shipped or imported by the
product or by the tests."""


def grad(*args, **kwargs):
    raise NotImplementedError("autograd is not available; the shim has no AD")
