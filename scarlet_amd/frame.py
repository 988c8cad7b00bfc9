"""Frames: shape, channels, PSF, WCS and dtype of a model or an observation
(reference scarlet/frame.py:9-287).  A WCS is anything with the small interface of
``scarlet_amd.wcs.LinearWCS`` (an ``astropy.wcs.WCS`` has it too)."""

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
        self.wcs = wcs
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
        """(y, x) pixel coordinate of a sky coordinate; without WCS they coincide."""
        sky = np.array(sky_coord, dtype=np.float64).reshape(-1, 2)
        if self.wcs is not None:
            pixel = np.array(self.wcs.celestial.world_to_pixel_values(sky)).reshape(-1, 2)
            sky = np.flip(pixel, axis=-1)  # FITS (x, y) -> (y, x)
        return sky[0] if sky.size == 2 else sky

    def get_sky_coord(self, pixel):
        pix = np.array(pixel, dtype=np.float64).reshape(-1, 2)
        if self.wcs is not None:
            pix = np.array(self.wcs.celestial.pixel_to_world_values(np.flip(pix, axis=-1)))
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
        """Common model frame of a set of observations (frame.py:155-287): the channels
        of all of them; the pixel grid of the reference observation (``obs_id``, default
        the finest one); its PSF (default: the narrowest, resampled to the model pixels
        if necessary); the union or intersection of the footprints, padded by half the
        widest PSF.  Every observation is matched to the new frame."""
        from . import interpolation

        assert coverage in ["union", "intersection"]
        if not hasattr(observations, "__iter__"):
            observations = (observations,)
        if all(o.wcs is None for o in observations) and model_wcs is None:
            return Frame._from_same_grid(observations, model_psf)
        scales, channels = [], []
        widest = narrowest = None
        for c, obs in enumerate(observations):
            channels = channels + list(obs.channels)
            h_obs = interpolation.get_pixel_size(interpolation.get_affine(obs.wcs))
            scales.append(h_obs)
            for psf in obs.psf.get_model():
                size = interpolation.get_psf_size(psf) * h_obs
                if widest is None or size > widest:
                    widest = size
                if (obs_id is None or c == obs_id) and model_psf is None and (
                        narrowest is None or size < narrowest):
                    narrowest = size
                    candidate, candidate_h = ImagePSF(psf[np.newaxis, :, :]), h_obs
        obs_ref = (observations[int(np.argmin(scales))] if obs_id is None
                   else observations[obs_id])
        if model_wcs is None:
            model_wcs = obs_ref.wcs
        h = interpolation.get_pixel_size(interpolation.get_affine(model_wcs))
        if model_psf is None:
            if candidate_h > h:
                angle, _ = interpolation.get_angles(model_wcs, obs.wcs)
                model_psf = ImagePSF(interpolation.sinc_interp_inplace(
                    candidate.get_model(), candidate_h, h, angle))
            else:
                model_psf = candidate
        probe = Frame((len(channels), 0, 0), channels=channels, psf=model_psf, wcs=model_wcs)
        model_box = None
        for obs in observations:
            if probe.wcs is obs.wcs:
                box = obs_ref.bbox[-2:]
            else:
                coord = obs.convert_pixel_to(probe)
                lo = np.floor(coord.min(axis=0)).astype("int")
                hi = np.ceil(coord.max(axis=0)).astype("int")
                box = Box.from_bounds((lo[0], hi[0] + 1), (lo[1], hi[1] + 1))
            if model_box is None:
                model_box = box
            elif coverage == "union":
                model_box = model_box | box
            else:
                model_box = model_box & box
        pad = int(np.round(widest / h / 2))
        model_box = model_box - (pad, pad)
        model_box.shape = tuple(s + 2 * pad for s in model_box.shape)
        model_wcs = model_wcs.deepcopy()
        # as the reference does (frame.py:277): (y, x) origin subtracted from the
        # FITS-order (x, y) crpix; the boxes are square-padded symmetric in practice
        model_wcs.wcs.crpix -= model_box.origin
        model_wcs.array_shape = model_box.shape
        frame = Frame((len(channels), *model_box.shape), channels=channels, psf=model_psf,
                      wcs=model_wcs)
        for obs in observations:
            obs.match(frame)
        return frame

    @staticmethod
    def _from_same_grid(observations, model_psf):
        """Observations without WCS live on one pixel grid: concatenated channels over
        their common footprint."""
        shapes = {o.shape[-2:] for o in observations}
        if len(shapes) != 1:
            raise NotImplementedError("observations with different footprints need a WCS")
        if model_psf is None:
            raise ValueError("model_psf is required without WCS information")
        channels = [c for o in observations for c in o.channels]
        frame = Frame((len(channels),) + tuple(shapes.pop()), channels=channels, psf=model_psf)
        for o in observations:
            o.match(frame)
        return frame
