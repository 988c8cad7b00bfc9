"""Import the read-only reference under the conda interpreter (the one with astropy).

BUILD-CONTAINER TOOLING.  Shared by the golden generators that need real WCS objects:
NumPy aliases the reference still uses, the autograd / proxmin shims (but NOT the astropy
stand-in: astropy must be the real one), stubs for the two compiled extension modules.
``scarlet = conda_reference.load()``.
"""
import atexit
import os
import shutil
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))

# NumPy aliases the reference still uses
for name, target in (("asscalar", "asarray"), ("alen", "asarray"), ("msort", "sort"),
                     ("sometrue", "any"), ("alltrue", "all"), ("product", "prod"),
                     ("cumproduct", "cumprod"), ("round_", "round"), ("asfarray", "asarray")):
    if not hasattr(np, name):
        setattr(np, name, getattr(np, target))
for name, t in (("float", float), ("int", int), ("bool", bool), ("object", object),
                ("complex", complex), ("str", str)):
    if name not in np.__dict__:
        setattr(np, name, t)

# only the autograd / proxmin shims: astropy must be the real one
shim_dir = tempfile.mkdtemp()
atexit.register(shutil.rmtree, shim_dir, ignore_errors=True)
for pkg in ("autograd", "proxmin"):
    os.symlink(os.path.join(HERE, "shims", pkg), os.path.join(shim_dir, pkg))
sys.path[:0] = [shim_dir, REPO, "/root/reference"]
# the compiled sweep / filter of the reference cannot be built here: the oracle's C
# restatement stands in (as in load_reference.py); the detection module is not used
sys.path.insert(0, HERE)
from load_reference import _native_stub  # noqa: E402

sys.modules["scarlet.operators_pybind11"] = _native_stub()
sys.modules["scarlet.detect_pybind11"] = types.ModuleType("scarlet.detect_pybind11")
for f in ("get_footprints", "get_connected_pixels", "get_connected_multipeak"):
    setattr(sys.modules["scarlet.detect_pybind11"], f, None)


def load():
    import scarlet

    return scarlet

