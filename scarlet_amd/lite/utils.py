"""Small helpers of ``scarlet.lite`` (reference scarlet/lite/utils.py)."""

import numpy as np

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
