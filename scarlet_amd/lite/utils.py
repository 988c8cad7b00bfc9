"""Small helpers of ``scarlet.lite`` (reference scarlet/lite/utils.py)."""

import numpy as np

from ..bbox import overlapped_slices


def insert_image(image_box, sub_box, sub_image, fill=0, dtype=None):
    """Image of ``image_box`` filled with ``fill``, with ``sub_image`` (located at
    ``sub_box``) written into the overlap."""
    dtype = sub_image.dtype if dtype is None else dtype
    image = np.full(image_box.shape, fill, dtype=dtype) if fill != 0 else np.zeros(
        image_box.shape, dtype=dtype)
    dst, src = overlapped_slices(image_box, sub_box)
    image[dst] = sub_image[src]
    return image
