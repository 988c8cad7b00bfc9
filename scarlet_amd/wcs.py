"""A minimal celestial WCS for multi-resolution scenes.

The reference takes ``astropy.wcs.WCS`` objects and uses a small part of them: the
linear transformation (``wcs.wcs.pc``, ``cdelt``, ``crpix``, ``crval``), pixel <-> sky
conversion of the celestial axes, ``deepcopy`` and ``array_shape`` (frame.py:84-153,
235-287; interpolation.py:378-424).  astropy is not available here, so this class
provides exactly that surface for the scales scarlet works at: images of a few arcseconds
to arcminutes, where every zenithal projection (TAN, AIR, ...) is linear to ~1e-9 of a
pixel.  The sky coordinates are (ra, dec) in degrees, the flat-sky offset from ``crval``
being ``(d_ra cos(dec0), d_dec) = cdelt * pc @ (pixel - (crpix - 1))`` with FITS-order
(x, y) pixels.  An ``astropy.wcs.WCS`` can be converted with ``LinearWCS.from_astropy``.
"""

import copy

import numpy as np


class _Params:
    """The ``wcs.wcs`` namespace: pc, cdelt, crpix (1-based, FITS order), crval."""

    def __init__(self, crpix, crval, pc, cdelt):
        self.crpix = np.array(crpix, dtype=np.float64)
        self.crval = np.array(crval, dtype=np.float64)
        self.pc = np.array(pc, dtype=np.float64)
        self.cdelt = np.array(cdelt, dtype=np.float64)


class LinearWCS:
    def __init__(self, crpix, crval, pc, cdelt=(1.0, 1.0), array_shape=None):
        self.wcs = _Params(crpix, crval, pc, cdelt)
        self.array_shape = array_shape

    @staticmethod
    def from_astropy(wcs):
        w = wcs.celestial.wcs
        return LinearWCS(w.crpix, w.crval, w.pc, w.cdelt, array_shape=wcs.array_shape)

    @property
    def celestial(self):
        return self

    def deepcopy(self):
        return copy.deepcopy(self)

    def _matrix(self):
        return self.wcs.cdelt[:, None] * self.wcs.pc

    def pixel_to_world_values(self, pix):
        """(x, y) zero-based pixels (n, 2) -> (ra, dec) in degrees (n, 2)."""
        pix = np.asarray(pix, dtype=np.float64).reshape(-1, 2)
        inter = (pix - (self.wcs.crpix - 1)) @ self._matrix().T
        cosd = np.cos(np.deg2rad(self.wcs.crval[1]))
        return np.stack((self.wcs.crval[0] + inter[:, 0] / cosd, self.wcs.crval[1] + inter[:, 1]),
                        axis=1)

    def world_to_pixel_values(self, sky):
        sky = np.asarray(sky, dtype=np.float64).reshape(-1, 2)
        cosd = np.cos(np.deg2rad(self.wcs.crval[1]))
        inter = np.stack(((sky[:, 0] - self.wcs.crval[0]) * cosd, sky[:, 1] - self.wcs.crval[1]),
                         axis=1)
        return inter @ np.linalg.inv(self._matrix()).T + (self.wcs.crpix - 1)


class TanWCS(LinearWCS):
    """Gnomonic (``RA---TAN`` / ``DEC--TAN``) projection with the same surface.

    Needed when the reference pixel lies far outside the image -- cut-outs of large
    mosaics keep the mosaic's ``crpix`` (the multi-resolution tutorial's HST cut-out has
    it 30 000 pixels away), where the flat-sky form above is off by a tenth of a pixel.
    Standard FITS convention (Calabretta & Greisen 2002, zenithal case, ``LONPOLE`` 180
    degrees): intermediate world coordinates ``(x, y) = cdelt * pc @ (pixel - (crpix - 1))``
    in degrees are the standard coordinates ``(xi, eta)`` of the tangent plane at ``crval``.
    """

    @staticmethod
    def from_header(header):
        """From a mapping with the FITS keywords of the two celestial axes (``CRPIX1/2``,
        ``CRVAL1/2`` and either ``CD1_1`` ... or ``PC1_1`` ... with ``CDELT1/2``)."""
        crpix = (header["CRPIX1"], header["CRPIX2"])
        crval = (header["CRVAL1"], header["CRVAL2"])
        prefix = "CD" if "CD1_1" in header else "PC"
        matrix = [[header.get("%s%d_%d" % (prefix, i, j), float(i == j) if prefix == "PC" else 0.0)
                   for j in (1, 2)] for i in (1, 2)]
        cdelt = (1.0, 1.0) if prefix == "CD" else (header.get("CDELT1", 1.0), header.get("CDELT2", 1.0))
        shape = (header["NAXIS2"], header["NAXIS1"]) if "NAXIS1" in header else None
        return TanWCS(crpix, crval, matrix, cdelt, array_shape=shape)

    @staticmethod
    def from_astropy(wcs):
        w = wcs.celestial.wcs
        return TanWCS(w.crpix, w.crval, w.get_pc(), w.get_cdelt(), array_shape=wcs.array_shape)

    def pixel_to_world_values(self, pix):
        pix = np.asarray(pix, dtype=np.float64).reshape(-1, 2)
        xi, eta = np.deg2rad((pix - (self.wcs.crpix - 1)) @ self._matrix().T).T
        ra0, dec0 = np.deg2rad(self.wcs.crval)
        along = np.cos(dec0) - eta * np.sin(dec0)
        ra = ra0 + np.arctan2(xi, along)
        dec = np.arctan2(np.sin(dec0) + eta * np.cos(dec0), np.hypot(xi, along))
        return np.stack((np.rad2deg(ra) % 360.0, np.rad2deg(dec)), axis=1)

    def world_to_pixel_values(self, sky):
        sky = np.asarray(sky, dtype=np.float64).reshape(-1, 2)
        ra, dec = np.deg2rad(sky).T
        ra0, dec0 = np.deg2rad(self.wcs.crval)
        d_ra = ra - ra0
        toward = np.sin(dec) * np.sin(dec0) + np.cos(dec) * np.cos(dec0) * np.cos(d_ra)
        xi = np.cos(dec) * np.sin(d_ra) / toward
        eta = (np.sin(dec) * np.cos(dec0) - np.cos(dec) * np.sin(dec0) * np.cos(d_ra)) / toward
        inter = np.rad2deg(np.stack((xi, eta), axis=1))
        return inter @ np.linalg.inv(self._matrix()).T + (self.wcs.crpix - 1)
