"""BASELINE config 5 (multi-resolution, reference tests/test_multiresolution.py and
docs/tutorials/multiresolution.ipynb) for ``bench.py --config cfg5``:

* forward rendering of a high-resolution image into a low-resolution observation
  (``ResolutionRenderer``; SURVEY.md 8d: "forward render only"): the fixture pair the
  survey timed on the reference (131^2 -> 78^2, 128 ms per render on one CPU core), device
  time through transforms along x (the operator's transforms read once: HBM roofline) and,
  beside it, as the two dense products on the matrix cores against the f32 MFMA peak;
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
    from scarlet_amd import _lib
    Fy, Fx = r._fft_shape
    Kx = Fx // 2 + 1
    ref = g["rendered_%d_%d_union" % (i, j)]
    err = float(np.abs(rendered - ref).max() / np.abs(ref).max())
    path = r.device_path()
    ms = {}
    for p in ((1, 0) if path == 1 else (0,)):  # the dense products beside the spectral path
        r.device_path(p)
        obs_lr.render(g["image_%d" % i][None])
        t = ctypes.c_double()
        _lib.check(lib.smi_resampler_time(handle, n_rep, ctypes.byref(t)))
        ms[p] = t.value
    r.device_path(path)
    # dense: two products per band.  spectral: the transform of the model rows as a product,
    # the contraction with the operator's transforms (8 flops per complex term) and the sum
    # over k; its HBM bytes are the operator's transforms, read once, + model in + rendering out
    flops = C * (2.0 * Fy * Fx * Fx * n_b + 2.0 * n_a * n_b * Fy * Fx)
    flops_spectral = C * (2.0 * Fy * Fx * 2 * Kx + 8.0 * n_a * Fy * Kx + 4.0 * n_a * n_b * Kx)
    bytes_spectral = C * (8.0 * n_a * Fy * Kx + 4.0 * Fy * Fx + 4.0 * n_a * n_b) + 8.0 * n_b * Kx
    return dict(path=path, ms=ms[path], ms_dense=ms[0], flops=flops, flops_spectral=flops_spectral,
                bytes_spectral=bytes_spectral, shape=(C, n_a, n_b, int(Fy), int(Fx)), err=err)


def tutorial_fit(scarlet, g, n_iter, path=None):
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
    if path is not None:
        obs_hsc.renderer.device_path(path)
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
                flops_per_iteration=flops, shape=(C, n_a, n_b, int(Fy), int(Fx)),
                path=r.device_path())


def main(args):
    import torch

    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    import scarlet_amd as scarlet

    golden = os.path.join(ROOT, "tests", "golden")
    pair = render_pair(scarlet, np.load(os.path.join(golden, "multiresolution.npz")), 0, 1,
                       max(args.steps, 10))
    tutorial = np.load(os.path.join(golden, "multires_tutorial.npz"))
    fit = tutorial_fit(scarlet, tutorial, args.steps)
    fit_dense = tutorial_fit(scarlet, tutorial, args.steps, path=0) if fit["path"] == 1 else fit
    tflops_dense = pair["flops"] / (pair["ms_dense"] * 1e-3) / 1e12
    dense = {
        "what": "the same rendering as two dense products per band on the matrix cores "
                "(smi_resampler_set_path(r, 0); the path of an operator that is not circulant)",
        "ms_per_render": round(pair["ms_dense"], 4), "flops_per_render": pair["flops"],
        "achieved": round(tflops_dense, 2), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": round(tflops_dense / F32_MFMA_PEAK_TFLOPS, 5),
        "fit_ms_per_step": round(1e3 * fit_dense["seconds"] / fit_dense["n"], 4),
        "fit_flops_per_iteration": fit["flops_per_iteration"],
    }
    if pair["path"] == 1:
        gbs = pair["bytes_spectral"] / (pair["ms"] * 1e-3) / 1e9
        roofline = {
            "bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s",
            "frac": round(gbs / 8000.0, 5), "traffic": None,
            "kernel": "spectral_forward_kernel (+ gemm_mfma_kernel for the transform of the model "
                      "rows): one ResolutionRenderer rendering through transforms along x -- the "
                      "operator's transforms are read once (%.1f MB); three dependent launches, "
                      "latency-bound at this size" % (pair["bytes_spectral"] / 1e6),
            "bytes_per_render": pair["bytes_spectral"], "flops_per_render": pair["flops_spectral"],
            "ms_per_render": round(pair["ms"], 4),
            "dense_equivalent_tflops": round(pair["flops"] / (pair["ms"] * 1e-3) / 1e12, 2),
            "measured": "HIP events around %d back-to-back renderings of the resident model"
                        % max(args.steps, 10),
            "dense_products": dense,
        }
    else:
        roofline = dict(dense, bound="mfma", traffic=None,
                        kernel="gemm_mfma_kernel (+ reduce_slices_kernel): one ResolutionRenderer "
                               "rendering = two batched products")
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
                        "C, n_a, n_b, Fy, Fx = %s) + 250x250 observation through the rocFFT "
                        "convolution (a frame beyond the fused kernel's 160^2); Blend.fit through the facade (host hook every 10 "
                        "iterations included)" % (fit["frame"], fit["shape"]),
            "iterations": fit["n"],
            "resampler_path": "spectral" if fit["path"] == 1 else "dense products",
            "render_pair": "fixture images 0 -> 1 (131^2 -> 78^2), operators %s, deviation "
                           "from the reference's rendering %.1e of the peak" % (pair["shape"], pair["err"]),
        },
        "roofline": roofline,
        "cpu_baseline": {
            "value": round(1e3 / 128.0, 2), "unit": "renders/s", "cores": 1, "kind": "reference",
            "sample": "the reference's own ResolutionRenderer.render on this pair, 128 ms on one "
                      "core of the build container (BASELINE.md section 1; not re-timed here: the "
                      "reference does not travel to the GPU box)",
        },
    }
    print(json.dumps(line), flush=True)
