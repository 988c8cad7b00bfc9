"""Small helpers of ``scarlet.lite`` (reference scarlet/lite/utils.py)."""

import numpy as np
from scipy.special import erfc

from ..bbox import Box, overlapped_slices
from ..initialization import get_minimal_boxsize


def insert_image(image_box, sub_box, sub_image, fill=0, dtype=None):
    """Image of ``image_box`` filled with ``fill``, with ``sub_image`` (located at
    ``sub_box``) written into the overlap."""
    dtype = sub_image.dtype if dtype is None else dtype
    image = np.full(image_box.shape, fill, dtype=dtype) if fill != 0 else np.zeros(
        image_box.shape, dtype=dtype)
    dst, src = overlapped_slices(image_box, sub_box)
    image[dst] = sub_image[src]
    return image


def bounds_to_bbox(bounds):
    """Box of inclusive ``(bottom, top, left, right)`` bounds (detect.py:15-26)."""
    return Box((bounds[1] + 1 - bounds[0], bounds[3] + 1 - bounds[2]), origin=(bounds[0], bounds[2]))


def project_morph_to_center(morph, center, bbox, fullbox, boxsize=None):
    """Cut a standard-size odd box centred on ``center`` out of the full-frame ``morph``
    whose support is ``bbox`` (lite/utils.py:41-103).  Returns ``(centered, box)``."""
    if bbox.contains(center):
        size = 2 * max(center[0] - bbox.start[-2], bbox.stop[0] - center[-2],
                       center[1] - bbox.start[-1], bbox.stop[1] - center[-1])
    else:
        size = 0
    if boxsize is None:
        boxsize = get_minimal_boxsize(size)
    half = boxsize // 2
    box = Box.from_bounds((center[0] - half, center[0] + half + 1),
                          (center[1] - half, center[1] + half + 1))
    centered = np.zeros(box.shape, dtype=morph.dtype)
    dst, src = overlapped_slices(box, fullbox)
    centered[dst] = morph[src]
    return centered, box


def integrated_gaussian(X, sigma):
    """1-D Gaussian integrated over the unit pixels centred on ``X``."""
    sqrt2 = np.sqrt(2)
    left = erfc((0.5 - X) / (sqrt2 * sigma))
    right = erfc((2 * X + 1) / (2 * sqrt2 * sigma))
    return np.sqrt(np.pi / 2) * sigma * (1 - left + 1 - right)


def integrated_circular_gaussian(X=None, Y=None, sigma=0.8):
    """Pixel-integrated circular Gaussian, unit sum: the usual model PSF (15 x 15 by
    default, lite/utils.py:127-156)."""
    if X is None:
        if Y is not None:
            raise Exception("Either X and Y must be specified, or neither must be specified, "
                            "got X={} and Y={}".format(X, Y))
        X = Y = np.arange(-7, 8)
    image = integrated_gaussian(X, sigma)[None, :] * integrated_gaussian(Y, sigma)[:, None]
    return image / np.sum(image)


def get_circle_mask(diameter, dtype=np.float64):
    """``(diameter, diameter)`` image that is 1 inside the inscribed circle, 0 outside
    (for even diameters centre and radius sit on the half pixel)."""
    c = (diameter - 1) / 2
    r = diameter / 2 if diameter % 2 == 0 else c
    x = np.arange(diameter)
    xx, yy = np.meshgrid(x, x)
    circle = np.ones((diameter, diameter), dtype=dtype)
    circle[np.sqrt((xx - c) ** 2 + (yy - c) ** 2) > r] = 0
    return circle
