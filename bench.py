"""Benchmark: PGM iterations/sec over batched blends (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A step = one proximal-gradient iteration (render -> FFT convolution -> weighted
residual/loss -> adjoint convolution -> gradient gather -> AMSGrad -> prox chain)
of EVERY blend of the rank's batch.  Workload = BASELINE.json configs[2]'s batch
of 1024 synthetic 5-band 128x128 blends with 10 ExtendedSource components each
(SURVEY.md section 8d), per GPU (weak scaling: rank r fits seeds
1234 + 1024 r + b).  Inputs are resident in HBM before the timed region.

Output: ONE JSON line on rank 0 with the contract's fields plus
  roofline     algorithmic bytes per blend-iteration x blend-iterations / device time
               of the timed region (HIP events on the batch stream), against 8 TB/s
  cpu_baseline the CPU oracle (NumPy/C port of the reference loop) timed on this
               host, single thread, on a bounded sample of the same workload.
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# SURVEY.md section 8d / BASELINE.md section 3: algorithmic bytes per blend-iteration
# (C=5, 128x128, K=10 boxes of 41x41, F=180x180 in the accounting):
#   B0    = 4 [2 N_pix + 8 P_el]            = 1 194 880   (N_pix = 81 920, P_el = 16 860)
#   B_fft = 4 transforms + 2 kernel reads   = 5 474 880
# split by kernel: the convolution kernel owns B_fft + data/weights + one parameter read,
# the update kernel owns the parameter write-back and the m/v/vhat read+write.
BYTES_FFT_PATH = 6_669_760
BYTES_NULL_PATH = 1_194_880
BYTES_CONV_KERNEL = 5_474_880 + 4 * (2 * 81_920 + 16_860)   # 6 197 680
BYTES_UPDATE_KERNEL = 4 * 7 * 16_860                        #   472 080
HBM_PEAK_GBS = 8000.0


def build_scenes(n_blends, seed0, device=0):
    """Synthetic scenes; the noiseless truth is rendered on the GPU."""
    from scarlet_amd import synthetic

    kern = synthetic.psfs()
    return kern, synthetic.make_batch(range(seed0, seed0 + n_blends), kernel=kern, device=device)


def cpu_baseline(scenes, n_blends, n_iter, e_rel):
    """Time the oracle (port of the reference loop) on the host: `n_blends` blends x
    `n_iter` iterations, one thread.  The oracle is only the thing being timed as the
    CPU baseline here; it is never part of the GPU path."""
    from oracle import pgm

    t0 = time.perf_counter()
    done = 0
    for s in scenes[:n_blends]:
        sc = pgm.Scene(
            s["data"].shape, s["data"], s["weights"], s["diff_kernel"],
            [pgm.Component(s["seds"][k].copy(), s["morphs"][k].copy(), s["origins"][k],
                           sed_min_step=s["noise_rms"]) for k in range(len(s["morphs"]))],
        )
        for it in range(n_iter):
            sc.step(it, e_rel)
            done += 1
    dt = time.perf_counter() - t0
    return done / dt, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--blends", type=int, default=1024, help="blends per GPU")
    ap.add_argument("--null-renderer", action="store_true", help="ablation: no PSF convolution")
    ap.add_argument("--fft", type=int, nargs=2, default=None, help="override FFT shape")
    ap.add_argument("--conv-path", default="auto", choices=["auto", "rocfft", "fused"])
    ap.add_argument("--cpu-blends", type=int, default=16)
    ap.add_argument("--cpu-iters", type=int, default=150)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--sub-ranges", type=int, default=0,
                    help="ranges of blends stepped on streams of their own (0 = library default)")
    ap.add_argument("--phases", action="store_true", help="print the per-phase device times")
    ap.add_argument("--loop", default="blend", choices=["blend", "lite-adaprox", "lite-fista"],
                    help="ablation: run the scarlet.lite loop (LiteBlend.fit semantics) on the "
                         "same scenes instead of Blend.fit's")
    args = ap.parse_args()

    import torch
    from scarlet_amd import BlendBatch, ComponentSpec, dist as sdist

    # SCARLET_AMD_DIST_BACKEND=gloo + SCARLET_AMD_SHARE_GPU=1 let several ranks share one GPU
    # (used only to exercise the multi-rank code path on a single-GPU box)
    backend = os.environ.get("SCARLET_AMD_DIST_BACKEND")
    share = os.environ.get("SCARLET_AMD_SHARE_GPU") == "1"
    if share:
        os.environ["LOCAL_RANK_REAL"] = os.environ.get("LOCAL_RANK", "0")
    rank, local_rank, world = sdist.env_rank()
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    if share:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    sdist.init_process_group(backend=backend, device_index=local_rank)
    assert world == args.gpus, "launch with torchrun --nproc-per-node == --gpus"
    e_rel = 1e-3  # Blend.fit default; tolerance of the prox sub-iterations

    nb = args.blends
    kern, scenes = build_scenes(nb, 1234 + rank * nb, device=local_rank)

    lite = args.loop != "blend"
    if lite:
        # LiteFactorizedComponent defaults (lite/models.py:143-180, lite/initialization.py:250-318)
        from scarlet_amd import _lib as slib

        flags = (slib.PROX_MONOTONIC | slib.PROX_FIT_CENTER | slib.PROX_CENTER_ON | slib.PROX_NORM_MAX)
        extra = (dict(fista_step=1.0 / (2 * 400.0)) if args.loop == "lite-fista" else {})
        comps = [
            [ComponentSpec(s["seds"][k], s["morphs"][k], s["origins"][k], prox_flags=flags,
                           sed_min_step=s["noise_rms"] / 10, center_floor=1e-20,
                           bg_level=np.full(5, 0.25 * s["noise_rms"], np.float32), **extra)
             for k in range(len(s["morphs"]))]
            for s in scenes
        ]
    else:
        comps = [
            [ComponentSpec(s["seds"][k], s["morphs"][k], s["origins"][k], sed_min_step=s["noise_rms"])
             for k in range(len(s["morphs"]))]
            for s in scenes
        ]
    data = np.stack([s["data"] for s in scenes])
    weights = np.stack([s["weights"] for s in scenes])
    # W warm-up + K timed iterations, then K more for the instrumented roofline pass
    total_it = args.warmup + 2 * args.steps
    batch = BlendBatch(
        data, weights, comps, kernel=None if args.null_renderer else kern[2],
        max_iter=total_it + 1, fft_shape=args.fft, device=local_rank, conv_path=args.conv_path,
        scheme="fista" if args.loop == "lite-fista" else "amsgrad", log_norm=not lite,
    )
    prox_max_iter = 1 if lite else 10  # lite applies the proximal operator once
    stream = torch.cuda.Stream(device=local_rank)
    batch.set_stream(stream.cuda_stream)
    batch.set_sub_ranges(args.sub_ranges)

    # warm-up iterations 0 .. W-1 (untimed), then K timed iterations of the same fit
    batch.step(0, args.warmup, e_rel=e_rel, prox_max_iter=prox_max_iter, check_convergence=False)
    torch.cuda.synchronize()
    sdist.barrier()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    batch.step(args.warmup, args.steps, e_rel=e_rel, prox_max_iter=prox_max_iter,
               check_convergence=False)
    ev1.record(stream)
    torch.cuda.synchronize()
    sdist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = sdist.max_over_ranks(elapsed)
    dev_ms = ev0.elapsed_time(ev1)

    active, err = batch.status()
    assert err < 0, "non-finite parameters in blend %d" % err
    loss = batch.loss_history()
    n_iter = np.array([len(l) for l in loss], dtype=np.int32)
    logL = np.array([-l[-1] for l in loss])
    n_iter_all, logL_all = sdist.gather_results(n_iter, logL)  # the only collective

    # Roofline pass: K further iterations of the same fit with HIP events around every
    # kernel, all blends in ONE range.  In the timed region above ranges of blends run on
    # streams of their own and their kernels share the chip, so the duration of a single
    # launch there says nothing about the kernel; `value` is not affected by this pass.
    ranges_timed = batch.sub_ranges()
    batch.set_sub_ranges(1)
    batch.enable_timing(True)  # HIP events around every phase, on the batch stream
    batch.step(args.warmup + args.steps, args.steps, e_rel=e_rel, prox_max_iter=prox_max_iter,
               check_convergence=False)
    torch.cuda.synchronize()
    phases = {k: round(v, 4) for k, v in batch.timing().items()}  # mean ms over the K steps
    batch.enable_timing(False)
    batch.set_sub_ranges(args.sub_ranges)

    if rank == 0:
        blend_iters = world * nb * args.steps
        value = blend_iters / elapsed
        bytes_per = BYTES_NULL_PATH if args.null_renderer else BYTES_FFT_PATH
        ms_iter = dev_ms / args.steps  # device time of one iteration of the whole batch
        fused = (not args.null_renderer) and phases["render"] < 0.05 * phases["conv"]
        if fused:
            # dominant kernel: fused_conv_kernel (render + conv + residual + conv^T)
            k_name, k_bytes, k_ms = "fused_conv_kernel", BYTES_CONV_KERNEL, phases["conv"]
        elif args.null_renderer:
            k_name, k_bytes, k_ms = "update_kernel_reg", BYTES_UPDATE_KERNEL, phases["update"]
        else:
            # rocFFT pipeline: no single dominant kernel of ours; price the whole iteration
            k_name, k_bytes, k_ms = "whole iteration (rocFFT pipeline)", bytes_per, ms_iter
        achieved = k_bytes * nb / (k_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as fh:
                rec = json.load(fh).get(k_name)
            if rec:  # measured with rocprofv3 --pmc (profiles/), bytes per launch
                traffic = rec["bytes_per_blend"] * nb
        roofline = {
            "bound": "hbm",
            "achieved": round(achieved, 2),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": traffic,
            "kernel": k_name,
            "algorithmic_bytes_per_launch": k_bytes * nb,
            "ms_per_launch": round(k_ms, 4),
            "measured": "HIP events on the batch stream over %d further iterations of the same "
                        "fit, one range of %d blends per launch (the timed region runs %d "
                        "ranges concurrently)" % (args.steps, nb, ranges_timed),
            "whole_iteration": {
                "algorithmic_bytes_per_blend_iteration": bytes_per,
                "ms": round(ms_iter, 4),
                "achieved": round(bytes_per * nb / (ms_iter * 1e-3) / 1e9, 2),
                "frac": round(bytes_per * nb / (ms_iter * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
            },
            "phases_ms": phases,
        }
        line = {
            "metric": "PGM iters/sec over batched blends; achieved HBM GB/s vs roofline",
            "value": round(value, 1),
            "unit": "blend-iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "configs[2] batch on one GPU: %d independent 5-band 128x128 blends "
                            "per GPU, 10 ExtendedSource components (41x41) each, %s" % (
                                nb, "NullRenderer ablation" if args.null_renderer
                                else "ConvolutionRenderer FFT %dx%d" % batch.fft_shape),
                "loop": args.loop,
                "blends_per_gpu": nb,
                "components_per_blend": 10,
                "parallelism": "blend-sharded x%d, no data-path collective" % world,
                "sub_ranges_per_gpu": ranges_timed,
                "mean_logL": float(np.mean(logL_all)),
            },
            "roofline": roofline,
        }
        if not args.no_cpu and not lite and world == 1:  # reported at N = 1 only
            v, dt = cpu_baseline(scenes, args.cpu_blends, args.cpu_iters, e_rel)
            line["cpu_baseline"] = {
                "value": round(v, 2),
                "unit": "blend-iterations/s",
                "cores": 1,
                "kind": "port",
                "sample": "%d blends x %d iterations of the same workload, NumPy/C oracle, "
                          "1 thread, %.1f s" % (args.cpu_blends, args.cpu_iters, dt),
            }
        print(json.dumps(line), flush=True)
    batch.close()
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
