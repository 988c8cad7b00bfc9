import numpy as _np


class VSpace:
    mappings = {_np.ndarray: None}

    @classmethod
    def register(cls, other, vspace_maker=None):
        return other
