"""Random scarlet.lite scenes (FISTA and adaprox parameters, centre-fitted monotonicity,
background threshold), GPU against the oracle's LiteScene: losses and parameters after a
few iterations.  Development aid.

    python tools/fuzz_lite.py [n_scenes] [seed]
"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import lite as olite  # noqa: E402
from scarlet_amd import _lib  # noqa: E402
from scarlet_amd.batch import BlendBatch, ComponentSpec  # noqa: E402
from test_gpu_parity import rel_err  # noqa: E402

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
worst = dict(loss=0.0, sed=0.0, morph=0.0)
bad = []
FLAGS = _lib.PROX_MONOTONIC | _lib.PROX_FIT_CENTER | _lib.PROX_CENTER_ON | _lib.PROX_NORM_MAX
for n in range(n_scenes):
    C = int(rng.integers(1, 7))
    H, W = int(rng.integers(30, 100)), int(rng.integers(30, 100))
    K = int(rng.integers(1, 7))
    kind = str(rng.choice(["fista", "adaprox"]))
    p = int(rng.choice([5, 9, 15]))
    yy, xx = np.mgrid[:p, :p] - p // 2
    sig = rng.uniform(0.8, 2.0, C)
    kernel = np.stack([np.exp(-(yy**2 + xx**2) / (2 * s**2)) for s in sig]).astype(np.float32)
    kernel /= kernel.sum(axis=(1, 2))[:, None, None]
    noise = rng.uniform(0.02, 0.1, C).astype(np.float32)
    truth = np.zeros((C, H, W), np.float32)
    specs, comps = [], []
    for k in range(K):
        h = w = int(rng.choice([11, 15, 21, 31]))
        oy, ox = int(rng.integers(0, max(H - h, 1))), int(rng.integers(0, max(W - w, 1)))
        y, x = np.mgrid[:h, :w]
        s = rng.uniform(1.2, 4.0)
        morph = np.exp(-((y - h // 2) ** 2 + (x - w // 2) ** 2) / (2 * s**2)).astype(np.float32)
        sed = rng.uniform(0.5, 3.0, C).astype(np.float32)
        ys, xs = slice(oy, min(oy + h, H)), slice(ox, min(ox + w, W))
        truth[:, ys, xs] += sed[:, None, None] * morph[None, : ys.stop - oy, : xs.stop - ox]
        start = (morph * rng.uniform(0.8, 1.2, morph.shape)).astype(np.float32)
        start /= start.max()
        sed0 = (sed * rng.uniform(0.7, 1.3, C)).astype(np.float32)
        step = float(1.0 / (2 * (1 / noise.mean() ** 2)))  # ~ 1 / Lipschitz of the weights
        extra = dict(fista_step=step) if kind == "fista" else dict(sed_min_step=noise / 10)
        specs.append(ComponentSpec(sed0, start, (oy, ox), prox_flags=FLAGS, neighbor_weight="angle",
                                   min_gradient=0.0, center_floor=1e-20, bg_level=noise * 0.25,
                                   morph_step=1e-2, **extra))
        okw = dict(fista_step=step) if kind == "fista" else dict(sed_min_step=noise / 10)
        comps.append(olite.LiteComponent(sed0.copy(), start.copy(), (oy, ox), noise, kind=kind, **okw))
    from oracle import fftconv
    images = (fftconv.convolve(truth, kernel, axes=(1, 2))
              + rng.normal(0, 1, truth.shape) * noise[:, None, None]).astype(np.float32)
    weights = np.broadcast_to((1 / noise**2)[:, None, None], truth.shape).astype(np.float32).copy()
    desc = "%s C=%d %dx%d K=%d p=%d" % (kind, C, H, W, K, p)
    scene = olite.LiteScene(images, weights, kernel, comps)
    batch = BlendBatch(images[None], weights[None], [specs], kernel=kernel, max_iter=10,
                       log_norm=False, scheme="fista" if kind == "fista" else "amsgrad")
    n_it = 6
    try:
        batch.step(0, n_it, e_rel=1e-6, prox_max_iter=1)
        loss = -np.array(batch.loss_history()[0])
        scene.fit(n_it, e_rel=0, resize=None)
        ref = np.array(scene.loss[:n_it])
        dev = dict(loss=np.abs(loss[:n_it] / ref - 1).max())
        seds, morphs = batch.parameters()
        dev["sed"] = max(rel_err(seds[k], c.sed) for k, c in enumerate(scene.components))
        dev["morph"] = max(np.abs(morphs[k] - c.morph).max() for k, c in enumerate(scene.components))
    finally:
        batch.close()
    for key, val in dev.items():
        worst[key] = max(worst[key], float(val))
    over = {k: float(v) for k, v in dev.items() if v > dict(loss=3e-4, sed=2e-3, morph=5e-3)[k]}
    if over:
        bad.append((n, desc, over))
    if os.environ.get("FUZZ_VERBOSE"):
        print(n, desc, {k: "%.1e" % v for k, v in dev.items()})
print("lite scenes: %d; worst deviations: %s" % (n_scenes, {k: "%.2e" % v for k, v in worst.items()}))
for entry in bad:
    print("OVER", entry)
