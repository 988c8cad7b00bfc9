from numpy import *  # noqa: F401,F403
from numpy import abs, max, min, sum, round, any, all, fft, random, linalg, ma  # noqa: F401
import numpy as _np

ndarray = _np.ndarray
float32 = _np.float32
float64 = _np.float64


def stack(arrays, *args, **kwargs):
    # scarlet/psf.py:117-124 passes a generator, which NumPy >= 1.24 rejects
    return _np.stack(list(arrays), *args, **kwargs)
