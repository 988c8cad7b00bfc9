"""Frames: shape, channels, PSF and dtype of a model or an observation
(reference scarlet/frame.py:9-153).  WCS-based multi-resolution frames
(``Frame.from_observations`` with different WCSs) are outside the scope of this
package: every frame here lives on one pixel grid."""

import logging

import numpy as np

from .bbox import Box
from .psf import PSF, ImagePSF

logger = logging.getLogger("scarlet_amd.frame")


class Frame:
    def __init__(self, shape, channels, wcs=None, psf=None, dtype=np.float32):
        self._bbox = Box(shape)
        assert len(channels) == self.C
        self.channels = channels
        if wcs is not None:
            raise NotImplementedError(
                "WCS frames (multi-resolution scenes) are not supported by scarlet_amd"
            )
        self.wcs = None
        if psf is None:
            logger.warning("No PSF specified. Possible, but dangerous!")
            self._psf = None
        else:
            self._psf = psf if isinstance(psf, PSF) else ImagePSF(np.asarray(psf))
        self.dtype = dtype

    @property
    def bbox(self):
        return self._bbox

    @property
    def shape(self):
        return self._bbox.shape

    @property
    def C(self):
        return self._bbox.shape[0]

    @property
    def Ny(self):
        return self._bbox.shape[1]

    @property
    def Nx(self):
        return self._bbox.shape[2]

    @property
    def psf(self):
        return self._psf

    def get_pixel(self, sky_coord):
        """Pixel coordinate of a sky coordinate; without WCS they coincide."""
        sky = np.array(sky_coord, dtype=np.float64).reshape(-1, 2)
        return sky[0] if sky.size == 2 else sky

    def get_sky_coord(self, pixel):
        pix = np.array(pixel, dtype=np.float64).reshape(-1, 2)
        return pix[0] if pix.size == 2 else pix

    def convert_pixel_to(self, target, pixel=None):
        """Pixel coordinates of this frame expressed in ``target``."""
        if pixel is None:
            y, x = np.indices(self.shape[-2:], dtype=np.float64)
            pixel = np.stack((y.flatten(), x.flatten()), axis=1)
        return target.get_pixel(self.get_sky_coord(pixel))

    @staticmethod
    def from_observations(observations, model_psf=None, model_wcs=None, obs_id=None,
                          coverage="union"):
        """Common model frame for observations on the same pixel grid: the
        concatenation of their channels over their common footprint."""
        if not hasattr(observations, "__iter__"):
            observations = (observations,)
        if model_wcs is not None or any(o.wcs is not None for o in observations):
            raise NotImplementedError("multi-resolution frames are not supported")
        shapes = {o.shape[-2:] for o in observations}
        if len(shapes) != 1:
            raise NotImplementedError("observations with different footprints")
        if model_psf is None:
            raise ValueError("model_psf is required without WCS information")
        channels = [c for o in observations for c in o.channels]
        frame = Frame((len(channels),) + tuple(shapes.pop()), channels=channels, psf=model_psf)
        for o in observations:
            o.match(frame)
        return frame
