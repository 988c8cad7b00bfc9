"""Random scene configurations, GPU against the oracle: forward (model, rendering, logL),
gradients and a few iterations of the loop.  Development aid: prints the worst deviations
and the configurations that exceed the tolerances of the parity tests.

    python tools/fuzz_vs_oracle.py [n_scenes] [seed]
"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pgm  # noqa: E402
from scarlet_amd import _lib  # noqa: E402
from scarlet_amd.batch import BlendBatch, ComponentSpec  # noqa: E402
from test_gpu_parity import grad_scales, rel_err  # noqa: E402

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
worst = dict(model=0, rendered=0, logL=0, g_sed=0, g_morph=0, chi=0, sed=0, morph=0)
bad = []
for n in range(n_scenes):
    C = int(rng.integers(1, 9))
    H, W = int(rng.integers(12, 150)), int(rng.integers(12, 150))
    K = int(rng.integers(1, 13))
    big = bool(os.environ.get("FUZZ_BIG"))  # boxes beyond the register-resident update kernels
    if big:
        H, W = int(rng.integers(90, 220)), int(rng.integers(90, 220))
        K = int(rng.integers(1, 5))
    null = rng.random() < 0.15
    per_band = rng.random() < 0.5
    p = int(rng.choice([3, 5, 9, 15, 21, 31, 41]))
    p = min(p, 2 * (min(H, W) // 2) - 1)
    kernel = None
    if not null:
        yy, xx = np.mgrid[:p, :p] - p // 2
        sig = rng.uniform(0.7, 2.5, C if per_band else 1)
        kernel = np.stack([np.exp(-(yy**2 + xx**2) / (2 * s**2)) for s in sig]).astype(np.float32)
        kernel /= kernel.sum(axis=(1, 2))[:, None, None]
    data = rng.normal(0, 1, (C, H, W)).astype(np.float32)
    weights = rng.uniform(0.5, 2.0, (C, H, W)).astype(np.float32)
    weights[rng.random((C, H, W)) < 0.05] = 0
    specs, comps = [], []
    for k in range(K):
        h, w = int(rng.integers(3, 62)), int(rng.integers(3, 62))
        if big:
            h, w = int(rng.integers(55, 126)), int(rng.integers(55, 126))
        oy, ox = int(rng.integers(-h // 2, H - h // 2)), int(rng.integers(-w // 2, W - w // 2))
        y, x = np.mgrid[:h, :w]
        s = rng.uniform(1.0, 6.0)
        morph = np.exp(-((y - h // 2) ** 2 + (x - w // 2) ** 2) / (2 * s**2))
        morph = (morph * rng.uniform(0.8, 1.2, morph.shape)).astype(np.float32)
        morph /= morph.max()
        sed = rng.uniform(0.2, 3.0, C).astype(np.float32)
        mode = str(rng.choice(["angle", "flat", "nearest"]))
        g = float(rng.choice([0.0, 0.1]))
        sym = bool(rng.random() < 0.3)
        if sym and rng.random() < 0.5:
            sym = float(rng.choice([0.25, 0.5, 0.9]))  # SymmetryConstraint(strength)
        flags = _lib.PROX_EXTENDED_SOURCE | (_lib.PROX_SYMMETRY if sym else 0)
        specs.append(ComponentSpec(sed, morph, (oy, ox), sed_min_step=0.01, prox_flags=flags,
                                   neighbor_weight=mode, min_gradient=g,
                                   sym_strength=1.0 if sym is True or not sym else sym))
        comps.append(pgm.Component(sed.copy(), morph.copy(), (oy, ox), sed_min_step=0.01,
                                   monotonic=mode, min_gradient=g, symmetric=sym))
    desc = "C=%d HxW=%dx%d K=%d kernel=%s" % (C, H, W, K, None if null else kernel.shape)
    if os.environ.get("FUZZ_LIST"):
        print(n, desc)
        continue
    only = os.environ.get("FUZZ_ONLY")
    if only and n not in [int(v) for v in only.split(",")]:
        continue
    if only:
        print(n, desc, [(c.morph.shape, c.origin, c.monotonic, c.symmetric) for c in comps])
    for path in (("auto",) if null else ("auto", "rocfft")):
        scene = pgm.Scene((C, H, W), data, weights, kernel, [
            pgm.Component(c.sed.copy(), c.morph.copy(), c.origin, sed_min_step=0.01,
                          monotonic=c.monotonic, min_gradient=c.min_gradient,
                          symmetric=c.symmetric) for c in comps])
        batch = BlendBatch(data[None], weights[None], [specs], kernel=kernel, max_iter=8,
                           conv_path=path)
        try:
            model, rendered, logL = batch.forward()
            ref_model = scene.get_model()
            ref_rendered = scene.render(ref_model)
            dev = dict(model=rel_err(model[0], ref_model), rendered=rel_err(rendered[0], ref_rendered))
            ref_logL = scene.log_likelihood(ref_rendered)
            dev["logL"] = abs(logL[0] - ref_logL) / abs(ref_logL)
            g_sed, g_morph = batch.gradient()
            _, grads = scene.loss_and_gradients()
            scene.loss = []
            dev["g_sed"] = max(np.abs(g_sed[k] - gs).max() / ss
                               for k, ((gs, gm), (ss, sm)) in enumerate(zip(grads, grad_scales(scene))))
            dev["g_morph"] = max(np.abs(g_morph[k] - gm).max() / sm
                                 for k, ((gs, gm), (ss, sm)) in enumerate(zip(grads, grad_scales(scene))))
            batch.step(0, 5, e_rel=1e-3)
            for it in range(5):
                scene.step(it, 1e-3)
            chi = np.array(batch.loss_history()[0]) - scene.log_norm
            chi_ref = np.array(scene.loss) - scene.log_norm
            dev["chi"] = np.abs(chi / chi_ref - 1).max()
            seds, morphs = batch.parameters()
            dev["sed"] = max(rel_err(seds[k], c.sed) for k, c in enumerate(scene.components))
            dev["morph"] = max(np.abs(morphs[k] - c.morph).max() for k, c in enumerate(scene.components))
        finally:
            batch.close()
        for key, val in dev.items():
            worst[key] = max(worst[key], float(val))
        limits = dict(model=1e-5, rendered=1e-5, logL=1e-5, g_sed=1e-5, g_morph=2e-5, chi=2e-4,
                      sed=2e-3, morph=5e-3)
        over = {k: float(v) for k, v in dev.items() if v > limits[k]}
        if over:
            bad.append((n, path, desc, over))
print("scenes: %d; worst deviations: %s" % (n_scenes, {k: "%.2e" % v for k, v in worst.items()}))
for entry in bad:
    print("OVER", entry)
