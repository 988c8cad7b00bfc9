"""Integer bounding boxes in model-frame coordinates.

Mirrors the public surface of the reference's ``scarlet/bbox.py`` (``Box`` at
bbox.py:4-276, ``overlapped_slices`` at bbox.py:279-301): 2-D boxes are
(height, width), 3-D boxes (channels, height, width); ``origin`` is the minimum
corner and may be negative (boxes are allowed to overhang the frame).
"""

import numpy as np


class Box:
    """Axis-aligned box ``[origin, origin + shape)`` in any dimension."""

    def __init__(self, shape, origin=None):
        self.shape = tuple(int(s) for s in shape)
        if origin is None:
            origin = (0,) * len(self.shape)
        if len(origin) != len(self.shape):
            raise AssertionError("origin and shape must have the same length")
        self.origin = tuple(int(o) for o in origin)

    # -- constructors ----------------------------------------------------
    @staticmethod
    def from_bounds(*bounds):
        """Box from (min, max) pairs, one per dimension; empty if max < min."""
        return Box(
            [max(0, hi - lo) for lo, hi in bounds], origin=[lo for lo, _ in bounds]
        )

    @staticmethod
    def from_data(X, min_value=0):
        """Tightest box around the elements of ``X`` above ``min_value``."""
        above = np.asarray(X) > min_value
        if not above.any():
            return Box.from_bounds(*([(0, 0)] * above.ndim))
        idx = np.nonzero(above)
        return Box.from_bounds(*[(int(i.min()), int(i.max()) + 1) for i in idx])

    # -- geometry --------------------------------------------------------
    @property
    def D(self):
        return len(self.shape)

    @property
    def start(self):
        return self.origin

    @property
    def stop(self):
        return tuple(o + s for o, s in zip(self.origin, self.shape))

    @property
    def center(self):
        return tuple(o + s / 2 for o, s in zip(self.origin, self.shape))

    @property
    def bounds(self):
        return tuple((o, o + s) for o, s in zip(self.origin, self.shape))

    @property
    def slices(self):
        return tuple(slice(o, o + s) for o, s in zip(self.origin, self.shape))

    def contains(self, p):
        if len(p) != self.D:
            raise ValueError(f"Dimension mismatch in {p} and {self.D}")
        return all(o <= q < o + s for q, o, s in zip(p, self.origin, self.shape))

    def grow(self, radius):
        if not hasattr(radius, "__iter__"):
            radius = [radius] * self.D
        return Box(
            [s + 2 * r for s, r in zip(self.shape, radius)],
            origin=[o - r for o, r in zip(self.origin, radius)],
        )

    # -- array access ----------------------------------------------------
    def extract_from(self, image, sub=None):
        """Copy the part of ``image`` under this box into ``sub`` (zeros where
        the box leaves the image)."""
        if sub is None:
            sub = np.zeros(self.shape, dtype=image.dtype)
        im_sl, sub_sl = overlapped_slices(Box(image.shape), self)
        sub[sub_sl] = image[im_sl]
        return sub

    def insert_into(self, image, sub):
        """Write ``sub`` into ``image`` at this box's position."""
        im_sl, sub_sl = overlapped_slices(Box(image.shape), self)
        image[im_sl] = sub[sub_sl]
        return image

    # -- algebra ---------------------------------------------------------
    def _check(self, other):
        if other.D != self.D:
            raise ValueError(f"Dimension mismatch in the boxes {other} and {self}")

    def __or__(self, other):
        self._check(other)
        return Box.from_bounds(
            *[
                (min(a0, b0), max(a1, b1))
                for (a0, a1), (b0, b1) in zip(self.bounds, other.bounds)
            ]
        )

    def __and__(self, other):
        self._check(other)
        return Box.from_bounds(
            *[
                (max(a0, b0), min(a1, b1))
                for (a0, a1), (b0, b1) in zip(self.bounds, other.bounds)
            ]
        )

    def __getitem__(self, i):
        shape, origin = self.shape[i], self.origin[i]
        if not hasattr(shape, "__iter__"):
            shape, origin = (shape,), (origin,)
        return Box(shape, origin=origin)

    def _offset(self, offset, sign):
        if not hasattr(offset, "__iter__"):
            offset = (offset,) * self.D
        return tuple(o + sign * int(d) for o, d in zip(self.origin, offset))

    def __iadd__(self, offset):
        self.origin = self._offset(offset, +1)
        return self

    def __add__(self, offset):
        return Box(self.shape, origin=self._offset(offset, +1))

    def __isub__(self, offset):
        self.origin = self._offset(offset, -1)
        return self

    def __sub__(self, offset):
        return Box(self.shape, origin=self._offset(offset, -1))

    def __matmul__(self, other):
        """Concatenate dimensions: (C,) @ (H, W) -> (C, H, W)."""
        return Box.from_bounds(*(self.bounds + other.bounds))

    def __imatmul__(self, other):
        return self.__matmul__(other)

    def copy(self):
        return Box(self.shape, origin=self.origin)

    __copy__ = copy

    def __eq__(self, other):
        return (
            isinstance(other, Box)
            and self.shape == other.shape
            and self.origin == other.origin
        )

    def __hash__(self):
        return hash((self.shape, self.origin))

    def __repr__(self):
        return "<Box shape={0}, origin={1}>".format(self.shape, self.origin)


def overlapped_slices(bbox1, bbox2):
    """Slices selecting the common region in an array spanning ``bbox1`` and in
    an array spanning ``bbox2`` (reference bbox.py:279-301)."""
    common = bbox1 & bbox2
    return (common - bbox1.origin).slices, (common - bbox2.origin).slices
