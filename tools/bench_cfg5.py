"""BASELINE config 5 (multi-resolution, reference tests/test_multiresolution.py and
docs/tutorials/multiresolution.ipynb) for ``bench.py --config cfg5``:

* forward rendering of a high-resolution image into a low-resolution observation
  (``ResolutionRenderer``; SURVEY.md 8d: "forward render only"): the fixture pair the
  survey timed on the reference (131^2 -> 78^2, 128 ms per render on one CPU core), device
  time of the two dense products on the matrix cores against the f32 MFMA peak;
* one fit of the tutorial scene (5-band 50^2 HSC cut-out + 250^2 HST cut-out, model frame
  6 x 282 x 282, four sources): iterations per second.

One JSON line like bench.py's; `value` is the tutorial fit's blend-iterations/s (one blend).
"""
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD


def render_pair(scarlet, g, i, j, n_rep):
    def wcs(k):
        w = scarlet.LinearWCS(g["crpix_%d" % k], g["crval_%d" % k], g["pc_%d" % k], g["cdelt_%d" % k])
        w.array_shape = g["crpix_%d" % k] * 2
        return w

    obs_hr = scarlet.Observation(g["image_%d" % i][None], wcs=wcs(i),
                                 psf=scarlet.ImagePSF(g["psf_%d" % i]), channels=["lr"])
    obs_lr = scarlet.Observation(g["image_%d" % j][None], wcs=wcs(j),
                                 psf=scarlet.ImagePSF(g["psf_%d" % j]), channels=["hr"])
    scarlet.Frame.from_observations([obs_lr, obs_hr], obs_id=1, coverage="union")
    r = obs_lr.renderer
    rendered = obs_lr.render(g["image_%d" % i][None])
    lib, handle, (C, n_a, n_b) = r._resampler()
    ms = ctypes.c_double()
    from scarlet_amd import _lib
    _lib.check(lib.smi_resampler_time(handle, n_rep, ctypes.byref(ms)))
    Fy, Fx = r._fft_shape
    flops = C * (2.0 * Fy * Fx * Fx * n_b + 2.0 * n_a * n_b * Fy * Fx)
    ref = g["rendered_%d_%d_union" % (i, j)]
    err = float(np.abs(rendered - ref).max() / np.abs(ref).max())
    return dict(ms=ms.value, flops=flops, shape=(C, n_a, n_b, int(Fy), int(Fx)), err=err)


def tutorial_fit(scarlet, g, n_iter):
    def wcs(tag, n):
        return scarlet.TanWCS(g["crpix_" + tag], g["crval_" + tag], g["pc_" + tag],
                              g["cdelt_" + tag], array_shape=(n, n))

    obs_hst = scarlet.Observation(g["data_hst"].copy(), wcs=wcs("hst", 250),
                                  psf=scarlet.ImagePSF(g["psf_hst"].copy()),
                                  channels=[str(c) for c in g["channels_hst"]])
    obs_hsc = scarlet.Observation(g["data_hsc"].copy(), wcs=wcs("hsc", 50),
                                  psf=scarlet.ImagePSF(g["psf_hsc"].copy()),
                                  channels=[str(c) for c in g["channels_hsc"]])
    observations = [obs_hsc, obs_hst]
    frame = scarlet.Frame.from_observations(observations, coverage="intersection",
                                            model_psf=scarlet.GaussianPSF(sigma=0.6))
    sources = [scarlet.ExtendedSource(frame, sky, observations, thresh=0.1)
               for sky in obs_hst.get_sky_coord(g["pixel_hst"])]
    scarlet.initialization.set_spectra_to_match(sources, observations)
    scarlet.Blend(sources, observations).fit(3, e_rel=1e-9)  # operators, plans, kernels resident
    blend = scarlet.Blend(sources, observations)
    t0 = time.perf_counter()
    n, logL = blend.fit(n_iter, e_rel=1e-12)
    dt = time.perf_counter() - t0
    r = obs_hsc.renderer
    _, _, (C, n_a, n_b) = r._resampler()
    Fy, Fx = r._fft_shape
    # per iteration: rendering + its transpose (two products each)
    flops = 2 * C * (2.0 * Fy * Fx * Fx * n_b + 2.0 * n_a * n_b * Fy * Fx)
    return dict(n=n, seconds=dt, frame=tuple(int(v) for v in frame.shape), logL=float(logL),
                flops_per_iteration=flops, shape=(C, n_a, n_b, int(Fy), int(Fx)))


def main(args):
    import torch

    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    import scarlet_amd as scarlet

    golden = os.path.join(ROOT, "tests", "golden")
    pair = render_pair(scarlet, np.load(os.path.join(golden, "multiresolution.npz")), 0, 1,
                       max(args.steps, 10))
    fit = tutorial_fit(scarlet, np.load(os.path.join(golden, "multires_tutorial.npz")), args.steps)
    tflops = pair["flops"] / (pair["ms"] * 1e-3) / 1e12
    line = {
        "metric": "PGM iters/sec over batched blends; achieved HBM GB/s vs roofline",
        "value": round(fit["n"] / fit["seconds"], 1),
        "unit": "blend-iterations/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * fit["seconds"] / fit["n"], 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
        "data": "fixture (tests/golden/multires_tutorial.npz, multiresolution.npz)",
        "config": {
            "workload": "configs[4]: multi-resolution tutorial scene, 1 blend, model frame %s, "
                        "5-band 50x50 observation through a ResolutionRenderer (operators "
                        "C, n_a, n_b, Fy, Fx = %s) + 250x250 observation through the fused "
                        "convolution; Blend.fit through the facade (host hook every 10 "
                        "iterations included)" % (fit["frame"], fit["shape"]),
            "iterations": fit["n"],
            "render_pair": "fixture images 0 -> 1 (131^2 -> 78^2), operators %s, deviation "
                           "from the reference's rendering %.1e of the peak" % (pair["shape"], pair["err"]),
        },
        "roofline": {
            "bound": "mfma", "achieved": round(tflops, 2), "peak": F32_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": round(tflops / F32_MFMA_PEAK_TFLOPS, 5), "traffic": None,
            "kernel": "gemm_mfma_kernel (+ reduce_slices_kernel): one ResolutionRenderer "
                      "rendering = two batched products",
            "flops_per_render": pair["flops"], "ms_per_render": round(pair["ms"], 4),
            "measured": "HIP events around %d back-to-back renderings of the resident model"
                        % max(args.steps, 10),
            "fit_flops_per_iteration": fit["flops_per_iteration"],
        },
        "cpu_baseline": {
            "value": round(1e3 / 128.0, 2), "unit": "renders/s", "cores": 1, "kind": "reference",
            "sample": "the reference's own ResolutionRenderer.render on this pair, 128 ms on one "
                      "core of the build container (BASELINE.md section 1; not re-timed here: the "
                      "reference does not travel to the GPU box)",
        },
    }
    print(json.dumps(line), flush=True)
