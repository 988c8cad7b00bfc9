"""Source initialisation (reference scarlet/initialization.py): runs once per
blend on the host before the fit.  The monotonicity sweep used here goes through
the C ABI (seam 1) to the GPU; rendering for the spectrum solve uses the device
renderer."""

import logging

import numpy as np

from .bbox import Box
from .morphology import get_minimal_boxsize  # noqa: F401  (reference exports it here)
from .renderer import ConvolutionRenderer, NullRenderer

logger = logging.getLogger("scarlet_amd.initialization")


def _as_tuple(observations):
    return observations if hasattr(observations, "__iter__") else (observations,)


def _warn_nonpositive(spectrum, sky_coord):
    if np.any(spectrum <= 0):
        msg = f"Zero or negative spectrum {spectrum} at {sky_coord}"
        (logger.warning if np.all(spectrum <= 0) else logger.info)(msg)


def get_pixel_spectrum(sky_coord, observations, correct_psf=False, models=None, concat=True):
    """Data values at the pixel of ``sky_coord`` in every channel; divided by
    the PSF peak (``correct_psf``) or by a model's value there (``models``)."""
    if models is not None:
        assert correct_psf is False
    if not hasattr(observations, "__iter__"):
        observations, models = (observations,), (models,)
    elif models is None:
        models = (None,) * len(observations)
    else:
        assert len(models) == len(observations)
    spectra = []
    for obs, model in zip(observations, models):
        iy, ix = np.round(obs.get_pixel(sky_coord)).astype("int")
        spectrum = obs.data[:, iy, ix].copy()
        if correct_psf and obs.psf is not None:
            spectrum /= obs.psf.get_model().max(axis=(1, 2))
        elif model is not None:
            spectrum /= model[:, iy, ix].copy()
        spectra.append(spectrum)
        _warn_nonpositive(spectrum, sky_coord)
    return np.concatenate(spectra).reshape(-1) if concat else spectra


def get_psf_spectrum(sky_coord, observations, compute_snr=False, concat=True):
    """PSF-weighted (matched-filter) flux per channel at ``sky_coord`` and,
    optionally, the signal-to-noise ratio of a point source there."""
    observations = _as_tuple(observations)
    spectra, num, den = [], [], []
    for obs in observations:
        index = np.round(obs.get_pixel(sky_coord)).astype("int")
        psf = obs.psf.get_model()
        bbox = obs.psf.bbox + (0, *index)
        img = bbox.extract_from(obs.data)
        noise = bbox.extract_from(obs.noise_rms)
        masked = bbox.extract_from(obs.noise_rms.mask)
        spec = []
        for c in range(obs.C):
            ok = ~masked[c]
            p, d = psf[c, ok], img[c, ok]
            flux = d @ p
            spec.append(flux / (p @ p))
            if compute_snr:
                num.append(flux)
                den.append((p * noise[c, ok] ** 2) @ p)
        spec = np.array(spec)
        spectra.append(spec)
        _warn_nonpositive(spec, sky_coord)
    if concat:
        spectra = np.concatenate(spectra).reshape(-1)
    if compute_snr:
        return spectra, np.sum(num) / np.sqrt(np.sum(den))
    return spectra


def trim_morphology(center_index, morph, bg_thresh=0, boxsize=None):
    """Zero the pixels at or below ``bg_thresh`` and cut a square box of
    standard size around ``center_index`` that holds what is left."""
    morph[~(morph > bg_thresh)] = 0
    bbox = Box.from_data(morph, min_value=0)
    if bbox.contains(center_index):
        size = 2 * max(
            center_index[0] - bbox.start[-2], bbox.stop[0] - center_index[-2],
            center_index[1] - bbox.start[-1], bbox.stop[1] - center_index[-1],
        )
    else:
        size = 0
    if boxsize is None:
        boxsize = get_minimal_boxsize(size)
    half = boxsize // 2
    bbox = Box.from_bounds(
        (center_index[0] - half, center_index[0] + half + 1),
        (center_index[1] - half, center_index[1] + half + 1),
    )
    return bbox.extract_from(morph), bbox


def build_initialization_image(observations, spectra=None):
    """Inverse-variance, spectrum-weighted coadd of all channels and its noise
    level: the detection image morphologies are initialised from."""
    if not hasattr(observations, "__iter__"):
        observations, spectra = (observations,), (spectra,)
    assert len(observations) == len(spectra)
    frame = observations[0].model_frame
    usable = [isinstance(o.renderer, (NullRenderer, ConvolutionRenderer)) for o in observations]
    if not hasattr(observations[0], "_detect"):
        detect, var = [], []
        for obs, ok in zip(observations, usable):
            if not ok:
                continue
            d = np.zeros(frame.shape, dtype=frame.dtype)
            v = np.zeros(frame.shape, dtype=frame.dtype)
            data_sl, model_sl = obs.renderer.slices
            obs.renderer.map_channels(d)[model_sl] += obs.data[data_sl]
            obs.renderer.map_channels(v)[model_sl] += obs.noise_rms[data_sl] ** 2
            detect.append(d)
            var.append(v)
        observations[0]._detect = (np.array(detect), np.array(var))
    detect, var = observations[0]._detect
    weights_c = []
    for obs, ok, spec in zip(observations, usable, spectra):
        if not ok:
            continue
        s = np.zeros(frame.C)
        obs.renderer.map_channels(s)[:] = 1 if spec is None else spec
        weights_c.append(s)
    spectrum = np.stack(weights_c, axis=0)[:, :, None, None]
    weight = np.zeros(var.shape)
    positive = var > 0
    weight[positive] = 1 / var[positive]
    weight *= spectrum
    return (weight * detect).sum(axis=(0, 1)), np.sqrt((spectrum * weight).sum(axis=(0, 1)))


def init_source(frame, center, observations, thresh=1, max_components=1, min_components=1,
                min_snr=50, shifting=False, resizing=True, boxsize=None, fallback=True):
    """One source at ``center`` with up to ``max_components`` components; with
    ``fallback`` the count is limited by the point-source SNR and reduced (down
    to a compact source) while initialisation yields non-finite parameters."""
    from .source import ExtendedSource

    observations = _as_tuple(observations)
    if fallback:
        _, snr = get_psf_spectrum(center, observations, compute_snr=True)
        by_snr = np.floor(snr / min_snr).astype("int")
        max_components = np.min([max_components, np.max([min_components, by_snr])])
    while max_components >= 0:
        try:
            if max_components > 0:
                source = ExtendedSource(frame, center, observations, thresh=thresh,
                                        shifting=shifting, resizing=resizing, boxsize=boxsize,
                                        K=max_components)
            else:
                source = ExtendedSource(frame, center, observations, shifting=shifting,
                                        resizing=resizing, boxsize=boxsize, compact=True)
            source.check_parameters()
        except ArithmeticError as exc:
            if not fallback:
                raise exc
            logger.info(f"Could not initialize source at {center} with {max_components} "
                        f"components: {exc}")
            max_components -= 1
            continue
        return source


# Detection images of a scene's sources, swept in one launch ahead of the loop over the sources
# (``prepare_detection_sweeps``): key -> (pixel spectra, coadd, coadd rms, symmetrised
# profile, the same profile made monotonic about the source's pixel, the trimming threshold the
# swept profile is good for -- None: any).  ``SingleExtendedSource`` looks its centre up here
# before it computes them itself.
_prepared = {}


def _prepared_key(sky_coord, observations):
    try:
        coord = tuple(float(v) for v in np.asarray(sky_coord, dtype=float).reshape(-1))
    except (TypeError, ValueError):
        return None
    return coord + tuple(id(obs) for obs in observations)


# Half-width of the window about a source's pixel in which its detection image is swept first
# (prepare_detection_sweeps); frames that are not much larger are swept whole.
SWEEP_WINDOW = 32


def _window(pixel, shape):
    R = SWEEP_WINDOW
    py, px = int(pixel[0]), int(pixel[1])
    return max(py - R, 0), min(py + R + 1, shape[0]), max(px - R, 0), min(px + R + 1, shape[1])


def _window_pays(shape, thresh):
    R = SWEEP_WINDOW
    return thresh is not None and thresh >= 0 and max(shape) > 2 * R + 1 + R // 2


def _rim_below(win, cut, shape, floor):
    """The largest value on the sides of the window that have frame beyond them is at or
    below ``floor`` (NaNs: no)."""
    y0, y1, x0, x1 = cut
    rim = [edge for beyond, edge in ((y0 > 0, win[0]), (y1 < shape[0], win[-1]),
                                     (x0 > 0, win[:, 0]), (x1 < shape[1], win[:, -1])) if beyond]
    return not rim or max(float(np.max(e)) for e in rim) <= floor


def sweep_in_window(profile, pixel, thresh, detect_std):
    """One image through the windowed sweep of ``prepare_detection_sweeps`` ('flat' weights,
    no minimal gradient): the swept profile with zeros outside the window, or None where the
    window does not pay or its rim is brighter than the smallest trimming threshold."""
    from . import operator

    if not _window_pays(profile.shape, thresh):
        return None
    if not (0 <= pixel[0] < profile.shape[0] and 0 <= pixel[1] < profile.shape[1]):
        return None
    cut = y0, y1, x0, x1 = _window(pixel, profile.shape)
    win = np.ascontiguousarray(profile[y0:y1, x0:x1])[None]
    operator.prox_weighted_monotonic_many(win, [(int(pixel[0]) - y0, int(pixel[1]) - x0)],
                                          neighbor_weight="flat", min_gradient=0)
    if not _rim_below(win[0], cut, profile.shape, thresh * float(np.min(detect_std))):
        return None
    full = np.zeros_like(profile)
    full[y0:y1, x0:x1] = win[0]
    return full


def prepare_detection_sweeps(frame, centers, observations, thresh=None):
    """What every ``SingleExtendedSource`` of the scene starts from -- the spectrum-weighted
    coadd about its centre, symmetrised and made monotonic (source.py:312-333) -- for ALL
    centres at once: the host part per centre, then ONE launch of the monotonic sweep for all
    of them (``operator.prox_weighted_monotonic_many``) instead of a host -> GPU -> host round
    trip per source.  Same arithmetic, same bits as the per-source path; centres the
    preparation cannot serve (outside the frame, a failing pixel spectrum) are left to it.

    With ``thresh`` (the caller's trimming threshold) the sweep runs in a window of
    ``2 SWEEP_WINDOW + 1`` pixels about the centre first.  The sweep bounds a pixel by the mean
    of its neighbours CLOSER to the centre ('flat' weights, no minimal gradient), and every
    closer neighbour of a pixel of a centred square lies in the square: inside the window the
    result is the full frame's, bit for bit, and every pixel outside is at most the largest
    value on the window's rim.  If that is at or below the smallest trimming threshold of the
    frame, ``trim_morphology`` zeroes everything outside whatever its value: the window's sweep
    IS the source's profile (tables of 65^2 instead of the frame's pixels, the same for every
    centre away from the border).  Otherwise the centre is swept on the whole frame."""
    from . import operator

    observations = _as_tuple(observations)
    _prepared.clear()
    rows = []
    for center in centers:
        key = _prepared_key(center, observations)
        if key is None or key in _prepared:
            continue
        try:
            per_obs = get_pixel_spectrum(center, observations, concat=False)
            coadd, coadd_rms = build_initialization_image(observations, spectra=per_obs)
            pixel = np.round(frame.get_pixel(center)).astype("int")
            if not (0 <= pixel[0] < coadd.shape[0] and 0 <= pixel[1] < coadd.shape[1]):
                continue
            profile = operator.prox_uncentered_symmetry(coadd.copy(), 0, center=pixel, algorithm="sdss")
        except Exception:  # the per-source path reports it
            continue
        rows.append((key, per_obs, coadd, coadd_rms, pixel, np.ascontiguousarray(profile)))
    if not rows:
        return 0

    def sweep(images_and_centres):
        """[(image, centre)] -> the swept images, one launch per image shape"""
        out = [None] * len(images_and_centres)
        by_shape = {}
        for n, (image, _) in enumerate(images_and_centres):
            by_shape.setdefault((image.shape, image.dtype.str), []).append(n)
        for group in by_shape.values():
            swept = np.stack([images_and_centres[n][0] for n in group])
            operator.prox_weighted_monotonic_many(
                swept, [tuple(int(v) for v in images_and_centres[n][1]) for n in group],
                neighbor_weight="flat", min_gradient=0)
            for n, image in zip(group, swept):
                out[n] = image
        return out

    shape = rows[0][5].shape
    whole = list(range(len(rows)))
    if _window_pays(shape, thresh):
        cuts = [_window(r[4], shape) for r in rows]
        swept = sweep([(np.ascontiguousarray(r[5][y0:y1, x0:x1]), (r[4][0] - y0, r[4][1] - x0))
                       for r, (y0, y1, x0, x1) in zip(rows, cuts)])
        whole = []
        for n, (r, cut, win) in enumerate(zip(rows, cuts, swept)):
            # (floor: the smallest trimming threshold of the frame)
            if not _rim_below(win, cut, shape, thresh * float(np.min(r[3]))):
                whole.append(n)
                continue
            full = np.zeros_like(r[5])
            full[cut[0]:cut[1], cut[2]:cut[3]] = win
            _prepared[r[0]] = (r[1], r[2], r[3], r[5], full, thresh)
    if whole:
        swept = sweep([(rows[n][5].copy(), rows[n][4]) for n in whole])
        for n, out in zip(whole, swept):
            r = rows[n]
            _prepared[r[0]] = (r[1], r[2], r[3], r[5], out, None)
    return len(rows)


def prepared_detection(sky_coord, observations):
    """The prepared images of a centre, or None."""
    if not _prepared:
        return None
    return _prepared.get(_prepared_key(sky_coord, _as_tuple(observations)))


def init_all_sources(frame, centers, observations, thresh=1, max_components=1,
                     min_components=1, min_snr=50, shifting=False, resizing=True, boxsize=None,
                     fallback=True, silent=False, set_spectra=True):
    """Initialise a source at every centre; returns ``(sources, skipped)``."""
    observations = _as_tuple(observations)
    sources, skipped = [], []
    centers = list(centers)
    try:
        # (all sources' detection images through the monotonic sweep in one launch)
        if max_components > 0 and len(centers) > 1:
            prepare_detection_sweeps(frame, centers, observations, thresh=thresh)
        return _init_all_sources(frame, centers, observations, thresh, max_components,
                                 min_components, min_snr, shifting, resizing, boxsize, fallback,
                                 silent, set_spectra)
    finally:
        _prepared.clear()


def _init_all_sources(frame, centers, observations, thresh, max_components, min_components,
                      min_snr, shifting, resizing, boxsize, fallback, silent, set_spectra):
    sources, skipped = [], []
    for k, center in enumerate(centers):
        try:
            sources.append(
                init_source(frame, center, observations, thresh=thresh,
                            max_components=max_components, min_components=min_components,
                            min_snr=min_snr, shifting=shifting, resizing=resizing,
                            boxsize=boxsize, fallback=fallback)
            )
        except Exception as exc:
            logger.warning(f"Failed to initialize source {k}")
            if not silent:
                raise exc
            skipped.append(k)
    if set_spectra:
        set_spectra_to_match(sources, observations)
    return sources, skipped


def set_spectra_to_match(sources, observations):
    """Best-fit amplitude of every component in every channel: weighted linear
    least squares of the rendered unit-spectrum component models against the data."""
    from .component import CombinedComponent

    observations = _as_tuple(observations)
    frame = observations[0].model_frame
    parameters, update_of, models, sums = [], [], [], []
    for i, src in enumerate(sources):
        comps = src.children if isinstance(src, CombinedComponent) else (src,)
        for j, comp in enumerate(comps):
            p = comp.get_parameter("spectrum")
            parameters.append(p)
            if p is not None and not p.fixed:
                p[:] = 1
            model = comp.get_model(frame=frame)
            target = len(models)
            # (np.allclose(a, b) implies |sum a - sum b| <= N atol + rtol sum |b|: most pairs are
            # told apart by two numbers instead of a pass over both cubes)
            total, bound = float(np.sum(model)), 1e-8 * model.size
            for prev, other in enumerate(models):
                gap = bound + 1e-5 * sums[prev][1]
                if abs(total - sums[prev][0]) > 2 * gap + 1e-9 * (abs(total) + abs(sums[prev][0])):
                    continue
                if np.allclose(model, other):
                    target = prev
                    logger.warning(
                        f"Source {i}, Component {j} has a model identical to another component.\n"
                        "This is likely not intended, and the source/component should be deleted. "
                        "Spectra will be identical.")
            update_of.append(target)
            if target == len(models):
                models.append(model)
                sums.append((total, float(np.sum(np.abs(model)))))
    models = np.array(models)
    n_models = len(models)
    for obs in observations:
        # The reference hands float64 unit-spectrum models to the renderer and solves the
        # normal equations in double (initialization.py:493-588); their condition numbers
        # are several hundred (250 .. 700 on the quickstart scene), so float32 convolutions
        # of the device would show up as 1e-4 in the spectra.  This one-off solve therefore
        # renders on the host in double where the renderer is a plain convolution.
        # (a subclass that overrides the rendering inherits render_float64 without meaning it)
        from .renderer import ConvolutionRenderer, NullRenderer

        precise = (obs.renderer.render_float64
                   if type(obs.renderer) in (ConvolutionRenderer, NullRenderer)
                   and hasattr(obs.renderer, "render_float64") else None)
        if precise is not None and not obs.parameters:
            rendered = np.stack([precise(m) for m in models], axis=0)
        else:
            rendered = np.stack([obs.render(m) for m in models], axis=0).astype(np.float64)
        spectra = np.zeros((n_models, obs.C))
        for c in range(obs.C):
            im = obs.data[c].reshape(-1).astype(np.float64)
            w = obs.weights[c].reshape(-1).astype(np.float64)
            m = rendered[:, c].reshape(n_models, -1)
            mw = m * w[None, :]
            seen = np.flatnonzero(np.sum(mw, axis=1) / np.sum(m, axis=1) / np.mean(w) > 0.1)
            if len(seen) == n_models:
                spectra[:, c] = np.linalg.inv(mw @ m.T) @ m @ (im * w)
            else:
                spectra[seen, c] = np.linalg.inv(mw[seen] @ m[seen].T) @ m[seen] @ (im * w)
        for k, p in enumerate(parameters):
            if p is not None and not p.fixed:
                obs.renderer.map_channels(p)[:] = spectra[update_of[k]]
    for p in parameters:
        if p is not None and p.constraint is not None:
            p[:] = p.constraint(p, 0)
