"""Post-fit measurements on components or model cubes (reference
scarlet/measure.py:6-149).  Host NumPy on downloaded models; nothing here is on
the fitting path."""

import numpy as np


def _model_and_origin(component):
    if hasattr(component, "get_model"):
        return component.get_model(), np.array(component.bbox.origin)
    return np.asarray(component), 0


def max_pixel(component):
    """Index (channel, y, x) of the brightest model pixel, in frame coordinates for
    a component."""
    model, origin = _model_and_origin(component)
    return tuple(np.array(np.unravel_index(np.argmax(model), model.shape)) + origin)


def flux(component):
    """Total flux per channel."""
    model, _ = _model_and_origin(component)
    return model.sum(axis=(1, 2))


def centroid(component):
    """Flux-weighted mean index along every axis (channel, y, x)."""
    model, origin = _model_and_origin(component)
    total = model.sum()
    grids = np.indices(model.shape)
    return np.array([(g * model).sum() for g in grids]) / total + origin


def snr(component, observations):
    """Matched-filter signal-to-noise with the rendered, flux-normalised model as
    the weight function: ``sum(M W) / sqrt(sum(var W^2))`` over all observations."""
    if not hasattr(observations, "__iter__"):
        observations = (observations,)
    if hasattr(component, "get_model"):
        model = component.get_model(frame=observations[0].model_frame)
    else:
        model = np.asarray(component)
    signal, weight, var = [], [], []
    for obs in observations:
        rendered = obs.render(model)
        signal.append(rendered.reshape(-1))
        weight.append((rendered / rendered.sum(axis=(-2, -1))[:, None, None]).reshape(-1))
        var.append((np.asarray(obs.noise_rms) ** 2).reshape(-1))
    signal, weight, var = map(np.concatenate, (signal, weight, var))
    return (signal * weight).sum() / np.sqrt((var * weight * weight).sum())


def moments(component, N=2, centroid=None, weight=None):
    """Image moments ``M[p, q] = sum (a0 - c0)^p (a1 - c1)^q model weight`` for all
    ``p + q <= N``, where a0 / a1 are the indices along the first / second spatial
    axis and ``centroid = (c1, c0)`` (the reference's convention, measure.py:132-139);
    per channel for a cube."""
    model, _ = _model_and_origin(component)
    if weight is None:
        weight = 1
    else:
        assert model.shape == np.shape(weight)
    if centroid is None:
        centroid = np.array(model.shape) // 2
    a0, a1 = np.indices(model.shape[-2:], dtype=np.float64)
    a1 = a1 - centroid[0]
    a0 = a0 - centroid[1]
    if model.ndim == 3:
        a0, a1 = a0[None], a1[None]
    return {
        (p, n - p): (a1**p * a0 ** (n - p) * model * weight).sum(axis=(-2, -1))
        for n in range(N + 1)
        for p in range(n + 1)
    }
