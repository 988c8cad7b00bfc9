"""Resampling helpers for multi-resolution rendering (reference
scarlet/interpolation.py:341-560, 708-739): WCS geometry, sinc interpolation of a PSF onto
a finer grid, the Fourier shift ramps.  Host-side set-up; nothing here runs per iteration.
"""

import numpy as np

from . import fft
from .fft import mk_shifter  # noqa: F401  (same name as in the reference module)


def get_affine(wcs):
    """Linear part of the WCS."""
    try:
        return wcs.wcs.pc
    except AttributeError:
        return wcs.cd


def get_pixel_size(model_affine):
    """Pixel scale of an affine transformation (interpolation.py:387-394)."""
    return np.sqrt(np.abs(model_affine[0, 0])
                   * np.abs(model_affine[1, 1] - model_affine[0, 1] * model_affine[1, 0]))


def get_angles(frame_wcs, model_wcs):
    """``([cos, sin], h)``: rotation between the two pixel grids and the ratio of their
    pixel scales (interpolation.py:397-424)."""
    model_affine, frame_affine = get_affine(model_wcs), get_affine(frame_wcs)
    model_pix, frame_pix = get_pixel_size(model_affine), get_pixel_size(frame_affine)
    h = frame_pix / model_pix
    u = np.sum(frame_affine, axis=0)[:2] / frame_pix
    w = np.sum(model_affine, axis=0)[:2] / model_pix
    u = u / np.sum(u**2) ** 0.5
    w = w / np.sum(w**2) ** 0.5
    return [np.dot(u, w), np.array(u[0] * w[1] - u[1] * w[0])], h


def get_psf_size(psf):
    """Rough 3-sigma radius in pixels from the area above half maximum
    (interpolation.py:708-739)."""
    above = psf / np.max(psf) > 0.5
    d = 2 * (np.sum(above) / np.pi) ** 0.5
    return 3 * d / (2 * (2 * np.log(2)) ** 0.5)


def sinc_interp(images, coord_hr, coord_lr, angle=None, padding=3):
    """Whittaker-Shannon interpolation of ``images`` sampled at ``coord_lr`` onto
    ``coord_hr`` (interpolation.py:427-502)."""
    y_hr, x_hr = coord_hr
    y_lr, x_lr = coord_lr
    hy, hx = np.abs(y_lr[1] - y_lr[0]), np.abs(x_lr[1] - x_lr[0])
    assert hy != 0 and hx != 0
    if (angle is None) or (1 - angle[0] < np.finfo(float).eps):
        sy = np.sinc((y_lr[np.newaxis, :] - y_hr[:, np.newaxis]) / hy)
        sx = np.sinc((x_lr[:, np.newaxis] - x_hr[np.newaxis, :]) / hx)
        return np.array([np.dot(np.dot(sy, image.T), sx) for image in images])
    # general case: every output row is read off a copy of the image that has been
    # Fourier-shifted by the rotated row offset, then sinc-interpolated along both axes
    # (interpolation.py:469-502; also taken for cos = 1 - 2e-16 from unrotated grids)
    cos, sin = angle
    fft_shape = fft._get_fft_shape(images, images, padding=padding, axes=[1, 2])
    X = fft.Fourier(images)
    X_fft = X.fft(fft_shape, (-2, -1))
    ramp_y, ramp_x = mk_shifter(fft_shape)
    shift_y = np.exp(ramp_y[np.newaxis, :] * (-(y_hr[:, np.newaxis]) * cos))
    shift_x = np.exp(ramp_x[np.newaxis, :] * (-(y_hr[:, np.newaxis]) * sin))
    result_fft = X_fft[:, np.newaxis, :, :] * shift_y[np.newaxis, :, :, np.newaxis]
    result_fft = result_fft * shift_x[np.newaxis, :, np.newaxis, :]
    result_shape = np.array([result_fft.shape[0], result_fft.shape[1], X.image.shape[1],
                             X.image.shape[2]])
    shifted = fft.Fourier.from_fft(result_fft, fft_shape, result_shape, [2, 3]).image
    shy = np.sinc((y_lr[np.newaxis, :] + x_hr[:, np.newaxis] * sin) / hy)
    shx = np.sinc((x_lr[np.newaxis, :] - x_hr[:, np.newaxis] * cos) / hx)
    result_y = (shifted[:, :, np.newaxis, :, :]
                * shy[np.newaxis, np.newaxis, :, :, np.newaxis]).sum(axis=-2)
    return (result_y * shx[np.newaxis, np.newaxis, :, :]).sum(axis=-1)


def sinc_interp_inplace(image, h_image, h_target, angle, pad_shape=None):
    """Interpolate a cube from pixel scale ``h_image`` to ``h_target`` over the same
    physical area, odd output size (interpolation.py:505-560)."""
    assert len(image.shape) == 3, "images should be provided as a cube"
    if pad_shape is not None:
        image = fft._pad(image, pad_shape, axes=[-2, -1])
    ny_lr, nx_lr = image.shape[-2:]
    coord_lr = np.array([np.arange(ny_lr) - (ny_lr - 1) / 2, np.arange(nx_lr) - (nx_lr - 1) / 2])
    ny_hr = int(np.round(image.shape[-2] * h_image / h_target))
    nx_hr = int(np.round(image.shape[-1] * h_image / h_target))
    ny_hr += ny_hr % 2 == 0
    nx_hr += nx_hr % 2 == 0
    coord_hr = np.array([np.arange(ny_hr) - (ny_hr - 1) / 2,
                         np.arange(nx_hr) - (nx_hr - 1) / 2]) / h_image * h_target
    return sinc_interp(image, coord_hr, coord_lr, angle=angle)
