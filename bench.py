"""Benchmark: PGM iterations/sec over batched blends (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config cfg3|cfg1|cfg4|cfg5] [--weak]

A step = one proximal-gradient iteration (render -> FFT convolution -> weighted
residual/loss -> adjoint convolution -> gradient gather -> AMSGrad -> prox chain)
of EVERY blend of the job.  The timed region is iterations 0 .. K-1 of a FRESH fit
(fixed-iteration fit from the initial parameters); the W warm-up
iterations run on the same batch beforehand and the parameters and optimizer state are
put back to their initial values before the clock starts, so the expensive early
iterations (up to 10 proximal sub-iterations each) are inside the timed region.

Workloads (BASELINE.json configs):
  cfg3 (default)  configs[2]: 1024 synthetic 5-band 128x128 blends IN TOTAL, 10
                  ExtendedSource components (41x41) each, seeds 1234 + b for global
                  blend b; rank r of N fits the contiguous shard
                  ``dist.shard_range(1024, r, N)`` (strong scaling, 128 blends per GPU
                  at N = 8).  At N = 1 this is configs[1]'s scene as a 1024-blend batch.
                  ``--weak`` gives every rank 1024 blends of its own instead.
  cfg1            configs[0]'s scene (tests/golden/hsc_cosmos_35.npz: 5x58x48, 10
                  components in boxes 21^2..61^2, per-band 43^2 difference kernel)
                  replicated into a batch.
  cfg4            configs[3]'s scene (tests/golden/point_source.npz = psf_unmatched_sim:
                  6x40x59, per-band 31^2 difference kernel, 3 PointSources + 2 ExtendedSources
                  in 71^2 / 81^2 boxes) replicated into a batch.
  cfg5            configs[4]: the multi-resolution operator (tools/bench_cfg5.py).

``python bench.py --gpus N`` without a torchrun environment starts its own N ranks
(``torch.distributed.run``, one per GPU; when the box has fewer GPUs than N the ranks
share GPUs over gloo -- a functional check, flagged in ``config``).

Output: ONE JSON line on rank 0 with the contract's fields plus
  roofline     dominant kernel: algorithmic bytes per launch / mean launch duration
               (HIP events on the batch stream over the same K iterations, replayed
               in one range of blends), at the FFT shape the kernel runs; `traffic`,
               `hbm_frac_measured` and `measured_hbm` from FETCH_SIZE / WRITE_SIZE passes of THIS
               run (the script re-runs itself under rocprofv3 for a few iterations; --no-counters
               or a missing profiler fall back to the committed profiles/hbm_traffic.json), VALU
               busy and the limiter from profiles/ (`counter_fields` lists what came from where); `bound` names the roofline priced against ("hbm"), `limiter`
               what the counters say limits the kernel ("unknown" without counters);
               `frac_physical` / `hbm_frac_whole_iteration`: the chip's physical utilisation
               over the whole iteration, to be read before the saturated `frac`; `speed_of_light`: the iteration against
               max(compulsory bytes / 8 TB/s, butterfly flops / f32 peak); `measured_hbm`: HBM
               bytes of the whole iteration by the PMC counters against the compulsory bytes
  parity       first and last blend of every rank's shard against the CPU oracle over the
               same K iterations (checker only, after the timed region)
  cpu_baseline the CPU oracle (NumPy/C port of the reference loop) timed on this host:
               one thread, and a process pool over all usable host cores; plus the
               reference's own forward timed beside the port's in the build container.
"""

import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0
F32_VECTOR_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: 256 CUs x 128 lanes x 2 flop x 2.4 GHz


def algorithmic_bytes(C, H, W, box_pixels, Fy, Fx, kernel_bands):
    """SURVEY.md 8d per blend-iteration, float32, with the FFT shape given.

    B0    = 4 [2 N_pix + P_el (read) + P_el (write) + 6 P_el (m, v, vhat r/w)]
    B_fft = 4 transforms x (real input 4 C Fy Fx + half spectrum 8 C Fy (Fx/2+1))
            + 2 reads of the kernel spectrum
    per kernel: the convolution kernel owns B_fft, data + weights and one parameter read;
    the update kernel the parameter write-back, the moments and its gather of the
    gradient image over every box (4 C N_k per component)."""
    n_pix = C * H * W
    p_el = sum(n + C for n in box_pixels)
    spec = 8 * Fy * (Fx // 2 + 1)
    b0 = 4 * (2 * n_pix + 8 * p_el)
    b_fft = 4 * (4 * C * Fy * Fx + C * spec) + 2 * kernel_bands * spec
    gather = 4 * C * sum(box_pixels)
    return dict(
        whole=b0 + b_fft,
        conv=b_fft + 4 * (2 * n_pix + p_el),
        update=4 * 7 * p_el + gather,
        null=b0,
        # what each kernel must move through HBM at ITS fusion boundary (the transforms of
        # the fused convolution never leave the CU): data + weights + parameters in, the
        # gradient image out; the update kernel's bytes are compulsory as they stand
        conv_boundary=4 * (3 * n_pix + p_el),
        update_boundary=4 * 7 * p_el + gather,
    )


def fft_flops(C, Fy, Fx):
    """Real-input 2-D transforms of one blend-iteration: 4 x C x 2.5 N log2 N."""
    n = Fy * Fx
    return 4 * C * 2.5 * n * np.log2(n)


# ----------------------------------------------------------------- CPU baseline
def _oracle_fit(args):
    """`n_iter` oracle iterations of one scene (worker of the CPU baseline)."""
    kind, seed, n_iter, e_rel = args
    from oracle import pgm

    if kind == "cfg1":
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from conftest import golden, hsc_scene

        sc = hsc_scene(golden("hsc_cosmos_35"))
    elif kind == "cfg4":
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from conftest import golden, point_scene

        sc = point_scene(golden("point_source"))
    else:
        from scarlet_amd import synthetic

        s = synthetic.make_blend(seed)
        sc = pgm.Scene(
            s["data"].shape, s["data"], s["weights"], s["diff_kernel"],
            [pgm.Component(s["seds"][k].copy(), s["morphs"][k].copy(), s["origins"][k],
                           sed_min_step=s["noise_rms"]) for k in range(len(s["morphs"]))])
    t0 = time.perf_counter()
    for it in range(n_iter):
        sc.step(it, e_rel)
    return n_iter, time.perf_counter() - t0


def oracle_check(scenes, picks, loss, K, e_rel):
    """Checker, outside the timed region: the loss histories the device produced for the
    blends `picks` of this rank against the CPU oracle run on the same scenes for the same
    K iterations.  Tolerances of tests/test_gpu_parity.py (relative, on the chi^2 part of
    the loss): 2e-5 over the first twelve iterations, 5e-4 afterwards (the transient of the
    whole-fit tests).  Returns the worst relative differences; raises if one is out of bounds."""
    from oracle import pgm

    worst_early, worst = 0.0, 0.0
    for i in picks:
        s = scenes[i]
        sc = pgm.Scene(
            s["data"].shape, s["data"], s["weights"], s["diff_kernel"],
            [pgm.Component(s["seds"][k].copy(), s["morphs"][k].copy(), s["origins"][k],
                           sed_min_step=s["noise_rms"]) for k in range(len(s["morphs"]))])
        for it in range(K):
            sc.step(it, e_rel)
        a = np.asarray(loss[i][:K], dtype=np.float64) - sc.log_norm
        b = np.asarray(sc.loss, dtype=np.float64) - sc.log_norm
        rel = np.abs(a - b) / np.abs(b)
        worst_early = max(worst_early, float(rel[:12].max()))
        worst = max(worst, float(rel.max()))
    if worst_early > 2e-5 or worst > 5e-4:
        raise AssertionError("device fit differs from the oracle: %.3g (first 12 iterations), "
                             "%.3g (all)" % (worst_early, worst))
    return worst_early, worst


def usable_cores():
    """Hardware threads this process may use: the affinity mask, capped by the cgroup
    CPU quota (a container sees all of the host's threads in os.cpu_count())."""
    n = len(os.sched_getaffinity(0))
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, period = fh.read().split()
            if q != "max":
                quota = float(q) / float(period)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    return n, quota


def cpu_baseline(kind, n_blends, n_iter, e_rel, pool_seconds=12.0):
    """The oracle (port of the reference loop) on the host, same workload from it = 0:
    (a) one thread, >= `n_blends` blends x `n_iter` iterations (about 8 s); (b) one worker process per
    usable core, every worker fitting whole blends (`OMP_NUM_THREADS=1`).  The oracle is
    only the thing being timed here; it is never part of the GPU path."""
    import multiprocessing as mp

    done, busy = 0, 0.0
    t0 = time.perf_counter()
    b = 0
    while b < n_blends or (busy < 8.0 and b < 512):  # at least `n_blends`, about 8 s
        n, dt = _oracle_fit((kind, 1234 + b, n_iter, e_rel))
        done += n
        busy += dt
        b += 1
    n_blends = b
    wall1 = time.perf_counter() - t0
    single = done / busy  # iteration loop only (scene construction excluded)

    n_aff, quota = usable_cores()
    workers = max(1, min(n_aff, int(quota) if quota else n_aff))
    # a profiler's preload must not follow into the pool's processes
    scrub = {k: os.environ.pop(k) for k in list(os.environ)
             if k in ("LD_PRELOAD", "HSA_TOOLS_LIB") or k.startswith(("ROCP", "ROCPROF"))}
    threads = {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS",
                                               "MKL_NUM_THREADS")}
    for k in threads:
        os.environ[k] = "1"
    pool_rate, pool_note = None, ""
    try:
        with mp.get_context("spawn").Pool(workers) as pool:
            pool.map(_oracle_fit, [(kind, 1234, 2, e_rel)] * workers)  # imports, library loads
            per_worker = max(1, int(round(pool_seconds * single / n_iter)))
            jobs = [(kind, 5000 + j, n_iter, e_rel) for j in range(workers * per_worker)]
            t0 = time.perf_counter()
            res = pool.map(_oracle_fit, jobs, chunksize=per_worker)
            wall = time.perf_counter() - t0
        its, busy_pool = sum(r[0] for r in res), sum(r[1] for r in res)
        # all workers are busy for the whole map (equal shares); the iteration loops alone
        # (scene construction excluded, as for the lone thread) took busy_pool / workers
        pool_rate = its / (busy_pool / workers)
        pool_note = ("%d worker processes x %d blends x %d iterations, %.1f s wall incl. scene "
                     "construction (%.0f/s by wall clock); per-worker rate %.2f of the lone "
                     "thread's" % (workers, per_worker, n_iter, wall, its / wall,
                                   its / busy_pool / single))
    finally:
        os.environ.update(scrub)
        for k, v in threads.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    out = {
        "value": round(pool_rate if pool_rate else single, 2),
        "unit": "blend-iterations/s",
        "cores": workers if pool_rate else 1,
        "kind": "port",
        "sample": "NumPy/C oracle, iterations 0..%d of the same workload; %s" % (
            n_iter - 1, pool_note),
        "single_thread": {
            "value": round(single, 2), "cores": 1,
            "sample": "%d blends x %d iterations, %.1f s" % (n_blends, n_iter, wall1)},
        "host": {"affinity_threads": n_aff, "cgroup_cpu_quota": quota,
                 "os_cpu_count": os.cpu_count()},
    }
    return out


# ------------------------------------------------------------------- workloads
def build_cfg3(lo, hi, device, lite_loop):
    """ComponentSpecs + observation of the synthetic blends with global indices lo..hi-1."""
    from scarlet_amd import ComponentSpec, synthetic

    kern = synthetic.psfs()
    scenes = synthetic.make_batch(range(1234 + lo, 1234 + hi), kernel=kern, device=device)
    if lite_loop:
        # LiteFactorizedComponent defaults (lite/models.py:143-180, lite/initialization.py:250-318)
        from scarlet_amd import _lib as slib

        flags = (slib.PROX_MONOTONIC | slib.PROX_FIT_CENTER | slib.PROX_CENTER_ON | slib.PROX_NORM_MAX)
        extra = (dict(fista_step=1.0 / (2 * 400.0)) if lite_loop == "lite-fista" else {})
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
    return data, weights, comps, kern[2], scenes


def build_cfg1(n):
    """`n` copies of the quickstart blend (initial sources as the reference's
    init_all_sources made them; golden fixture)."""
    from scarlet_amd import ComponentSpec

    g = np.load(os.path.join(ROOT, "tests", "golden", "hsc_cosmos_35.npz"))
    one = [ComponentSpec(g["sed_%d" % k], g["morph_%d" % k], g["origin_%d" % k],
                         sed_min_step=g["min_step_%d" % k]) for k in range(int(g["n_comp"]))]
    data = np.broadcast_to(g["images"], (n,) + g["images"].shape)
    weights = np.broadcast_to(g["weights"], (n,) + g["weights"].shape)
    return np.ascontiguousarray(data), np.ascontiguousarray(weights), [one] * n, g["diff_kernel"]


def build_cfg4(n):
    """`n` copies of the point-source tutorial blend on psf_unmatched_sim (sources as the
    reference initialised them; golden fixture): per-band difference kernel, three stars
    with free centres, two extended sources in boxes beyond the register-resident classes."""
    from scarlet_amd import ComponentSpec, PointSourceSpec

    g = np.load(os.path.join(ROOT, "tests", "golden", "point_source.npz"))
    one = []
    for k in range(int(g["n_src"])):
        if g["is_star"][k]:
            one.append(PointSourceSpec(g["sed_%d" % k], g["center_%d" % k], 0.9,
                                       sed_min_step=g["min_step_%d" % k]))
        else:
            one.append(ComponentSpec(g["sed_%d" % k], g["morph_%d" % k], g["origin_%d" % k],
                                     sed_min_step=g["min_step_%d" % k]))
    data = np.broadcast_to(g["images"], (n,) + g["images"].shape)
    weights = np.full(data.shape, 0.25, dtype=np.float32)
    return np.ascontiguousarray(data), weights, [one] * n, g["diff_kernel"]


def dominant_kernel(phases, conv_path, by):
    """(name, algorithmic bytes per blend, ms per launch) of the kernel the roofline object
    prices: the larger of the two kernels of a fused-path iteration -- the convolution kernel or
    the update phase -- by the event times of this run; the update kernel without a
    convolution; the whole iteration for the rocFFT pipeline, which has no single dominant
    kernel of ours.  `conv_path` is what the library says it runs (BlendBatch.conv_path), not
    something inferred from the times."""
    if conv_path == "none":
        return "update_kernel_reg", by["update"], phases["update"]
    if conv_path == "fused":
        if phases["update"] > phases["conv"]:
            return "update kernels (update phase of the iteration)", by["update"], phases["update"]
        return "fused_conv_kernel", by["conv"], phases["conv"]
    return "whole iteration (rocFFT pipeline)", by["whole"], phases["total"]


def build_facade_blends(lo, hi, device, times=None):
    """configs[2]'s scenes as the objects a scarlet script holds: one ``Blend`` per scene,
    a Frame / Observation pair matched by the difference kernel, ten ExtendedSource-style
    components (TabulatedSpectrum + ExtendedSourceMorphology, ``resizing=True`` -- the
    reference's default, source.py:215-260) from the same initial parameters as the C-ABI
    bench."""
    import scarlet_amd as scarlet
    from scarlet_amd import synthetic

    kern = synthetic.psfs()
    t0 = time.perf_counter()
    scenes = synthetic.make_batch(range(1234 + lo, 1234 + hi), kernel=kern, device=device)
    t1 = time.perf_counter()
    channels = list("grizy")
    model_psf = scarlet.GaussianPSF(sigma=(synthetic.SIGMA_MODEL,) * 5)
    obs_psf = scarlet.ImagePSF(np.repeat(kern[0], 5, axis=0))
    frame = scarlet.Frame((5, synthetic.H, synthetic.W), psf=model_psf, channels=channels)
    renderer = None
    blends = []
    for s in scenes:
        obs = scarlet.Observation(s["data"], psf=obs_psf, weights=s["weights"], channels=channels)
        # (one difference kernel for all: the PSFs are the same objects)
        obs.match(frame, renderer=renderer)
        if renderer is None:
            renderer = obs.renderer
        sources = []
        for k in range(len(s["morphs"])):
            oy, ox = (int(v) for v in s["origins"][k])
            h, w = s["morphs"][k].shape
            box = scarlet.Box((5, h, w), origin=(0, oy, ox))
            spectrum = scarlet.TabulatedSpectrum(frame, s["seds"][k].copy(), bbox=box[0],
                                                 min_step=s["noise_rms"])
            morphology = scarlet.ExtendedSourceMorphology(
                frame, (oy + h // 2, ox + w // 2), s["morphs"][k].copy(), bbox=box[1:],
                monotonic="angle", resizing=True)
            sources.append(scarlet.FactorizedComponent(frame, spectrum, morphology))
        blends.append(scarlet.Blend(sources, obs))
    if times is not None:  # the synthetic data (not a user's cost) apart from the objects
        times["scenes_s"] = t1 - t0
        times["objects_s"] = time.perf_counter() - t1
    return blends


def facade(args):
    """``--facade``: the path a scarlet user calls.  N Blend objects in, fit_blends(blends,
    K, e_rel=1e-4) with box resizing on, fitted Blend objects out; wall clock around the
    call, blend-iterations actually run / that time, beside the C-ABI rate of the same box
    for the same number of iterations (BlendBatch.step on device-resident inputs, no
    resizing: the headline measurement)."""
    import cProfile
    import pstats

    import torch
    import scarlet_amd as scarlet
    from scarlet_amd import BlendBatch, _lib

    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    n, K = args.blends, args.steps
    t0 = time.perf_counter()
    warm = build_facade_blends(0, min(n, 8), 0)
    scarlet.fit_blends(warm, 12, e_rel=1e-4)  # library load, plan cache, first launches
    times = {}
    blends = build_facade_blends(0, n, 0, times)
    t_build = time.perf_counter() - t0
    lib = _lib.load()
    uploads0 = lib.smi_observation_uploads()
    prof = cProfile.Profile() if args.profile else None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if prof:
        prof.enable()
    results = scarlet.fit_blends(blends, K, e_rel=1e-4)
    if prof:
        prof.disable()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    uploads = lib.smi_observation_uploads() - uploads0
    its = int(sum(r[0] for r in results))
    if prof:
        pstats.Stats(prof).sort_stats("cumulative").print_stats(45)
    # the C-ABI rate on the same scenes: same number of iterations per blend, no resizing
    data, weights, comps, kernel, _ = build_cfg3(0, n, 0, None)
    batch = BlendBatch(data, weights, comps, kernel=kernel, max_iter=K + 1)
    batch.save_state()
    batch.step(0, min(K, 10), e_rel=1e-3)
    batch.restore_state()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    batch.step(0, K, e_rel=1e-3)
    torch.cuda.synchronize()
    abi = n * K / (time.perf_counter() - t1)
    batch.close()
    resized = sum(1 for b in blends for src in b.sources
                  if tuple(src.children[1].bbox.shape) != (41, 41))
    line = {
        "metric": "PGM iters/sec over batched blends, through scarlet's Python API "
                  "(fit_blends: Blend objects in, fitted Blend objects out)",
        "value": round(its / elapsed, 1), "unit": "blend-iterations/s", "n_gpus": 1,
        "steps": K, "warmup": 0, "ms_per_step": round(elapsed / K * 1e3, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "configs[2] scenes as %d scarlet Blend objects (10 components each, "
                        "ExtendedSourceMorphology with resizing=True), fit_blends(blends, %d, "
                        "e_rel=1e-4); wall clock around the call" % (n, K),
            "blend_iterations": its, "wall_s": round(elapsed, 4),
            "iterations_per_blend": [int(min(r[0] for r in results)), int(max(r[0] for r in results))],
            "components_resized": resized,
            "observation_uploads_during_fit": int(uploads),
            "c_abi_rate_same_box": round(abi, 1),
            "ratio_to_c_abi": round(its / elapsed / abi, 4),
            "object_construction_s": round(times["objects_s"], 2),
            "synthetic_scenes_s": round(times["scenes_s"], 2),
            "setup_total_s": round(t_build, 2),
        },
    }
    print(json.dumps(line), flush=True)


def counters(kernel):
    """Counter-derived figures of the dominant kernel from the committed rocprofv3 PMC
    summaries (profiles/hbm_traffic.json, written by tools/hbm_counters.py)."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if not os.path.exists(path):
        return {}
    with open(path) as fh:
        return json.load(fh).get(kernel, {})


def live_hbm_counters(nb, timeout=90):
    """HBM bytes per blend of the two kernels of the iteration, measured now: this script run
    again under ``rocprofv3 --kernel-trace --pmc`` (FETCH_SIZE and WRITE_SIZE in passes of their
    own, a few iterations in one range of blends), reduced like tools/hbm_counters.py does --
    both counters are reported in KiB, FETCH_SIZE counts 64 B per 128-byte request on gfx950 and
    is doubled, the mean is taken over the full-batch launches -- and a third pass with
    SQ_ACTIVE_INST_VALU / SQ_ACTIVE_INST_ANY / GRBM_GUI_ACTIVE for the pipe utilisations.  Returns
    ``{kernel: {bytes_per_blend, valu_busy, issue_busy}}`` or ``None`` when the profiler is missing, fails or takes longer than ``timeout``
    seconds per pass (the line then falls back to the committed summary and says so)."""
    import csv
    import glob
    import shutil
    import tempfile

    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None
    # already under a profiler (rocprofv3 around this run): no profiler inside a profiler
    if any(k in ("HSA_TOOLS_LIB", "ROCP_TOOL_LIBRARIES") or k.startswith("ROCPROF") for k in os.environ):
        return None
    out = {}
    tmp = tempfile.mkdtemp(prefix="smi_pmc_", dir="/tmp")
    try:
        sums = {}
        passes = (("FETCH_SIZE",), ("WRITE_SIZE",),
                  ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE"))
        for n_pass, group in enumerate(passes):
            d = os.path.join(tmp, "pass%d" % n_pass)
            cmd = [prof, "--kernel-trace", "--pmc", *group, "-d", d, "-o", "run",
                   "--output-format", "csv", "--", sys.executable, os.path.abspath(__file__),
                   "--steps", "6", "--warmup", "2", "--no-cpu", "--sub-ranges", "1",
                   "--blends", str(nb), "--no-counters"]
            env = dict(os.environ, TMPDIR="/tmp", GPU_MAX_HW_QUEUES="8")
            res = subprocess.run(cmd, cwd="/tmp", env=env, timeout=timeout,
                                 stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            if res.returncode != 0:
                if n_pass < 2:
                    return None
                break  # (the byte counters stand without the utilisation pass)
            per_kernel = {}
            for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(path) as fh:
                    for row in csv.DictReader(fh):
                        if row["Counter_Name"] not in group:
                            continue
                        name = row["Kernel_Name"].replace("(anonymous namespace)::", "")
                        name = name.replace("void ", "").replace("smi::", "").split("<")[0].split("(")[0]
                        per_kernel.setdefault((name, row["Counter_Name"]), []).append(float(row["Counter_Value"]))
            for (name, counter), vals in per_kernel.items():
                if len(vals) < 6:
                    continue  # (not a kernel of the iteration)
                vals = sorted(vals)[len(vals) // 2:]  # the full-batch launches
                sums.setdefault(name, {})[counter] = sum(vals) / len(vals)
        for name, c in sums.items():
            if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                rec = {"bytes_per_blend": int(round((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 / nb))}
                # SQ_ACTIVE_INST_* count quad-cycles summed over the 1024 SIMDs, GRBM_GUI_ACTIVE
                # cycles summed over the 8 XCDs (tools/hbm_counters.py)
                if c.get("GRBM_GUI_ACTIVE") and "SQ_ACTIVE_INST_VALU" in c:
                    cycles = c["GRBM_GUI_ACTIVE"] / 8
                    rec["valu_busy"] = round(4 * c["SQ_ACTIVE_INST_VALU"] / (cycles * 1024), 4)
                    if "SQ_ACTIVE_INST_ANY" in c:
                        rec["issue_busy"] = round(4 * c["SQ_ACTIVE_INST_ANY"] / (cycles * 1024), 4)
                    rec["kernel_cycles"] = round(cycles)
                out[name] = rec
    except (subprocess.TimeoutExpired, OSError, KeyError, ValueError):
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out or None


def self_launch(args):
    """`python bench.py --gpus N` outside torchrun: start N ranks of this script."""
    import torch
    from scarlet_amd import dist as sdist

    env = dict(os.environ)
    if torch.cuda.device_count() < args.gpus:
        env["SCARLET_AMD_SHARE_GPU"] = "1"
        env["SCARLET_AMD_DIST_BACKEND"] = "gloo"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = sdist.launch_command(args.gpus, os.path.abspath(__file__), sys.argv[1:])
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="cfg3", choices=["cfg3", "cfg1", "cfg4", "cfg5"])
    ap.add_argument("--blends", type=int, default=1024,
                    help="blends of the whole job (per GPU with --weak)")
    ap.add_argument("--weak", action="store_true", help="--blends per GPU instead of in total")
    ap.add_argument("--null-renderer", action="store_true", help="ablation: no PSF convolution")
    ap.add_argument("--fft", type=int, nargs=2, default=None, help="override FFT shape")
    ap.add_argument("--conv-path", default="auto", choices=["auto", "rocfft", "fused"])
    ap.add_argument("--cpu-blends", type=int, default=12)
    ap.add_argument("--cpu-iters", type=int, default=0, help="0 = --steps")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--steady", action="store_true",
                    help="round-1 window: time iterations W .. W+K-1 of the warmed-up fit")
    ap.add_argument("--sub-ranges", type=int, default=0,
                    help="ranges of blends stepped on streams of their own (0 = library default)")
    ap.add_argument("--loop", default="blend", choices=["blend", "lite-adaprox", "lite-fista"],
                    help="ablation: run the scarlet.lite loop (LiteBlend.fit semantics) on the "
                         "same scenes instead of Blend.fit's")
    ap.add_argument("--ramp-ms", type=float, default=150.0,
                    help="milliseconds of untimed iterations before the warm-up (GPU clocks up "
                         "after the idle set-up; 0: none)")
    ap.add_argument("--no-counters", action="store_true",
                    help="do not re-run under rocprofv3 for the HBM byte counters (roofline.traffic "
                         "then comes from the committed profiles/hbm_traffic.json)")
    ap.add_argument("--facade", action="store_true",
                    help="time scarlet_amd.fit_blends on Blend objects built from the cfg3 "
                         "scenes (resizing on) instead of the C-ABI batch")
    ap.add_argument("--profile", action="store_true", help="--facade: cProfile of the call")
    args = ap.parse_args()

    # before the HIP runtime starts (scarlet_amd._lib explains): eight hardware queues, so that
    # a small shard can step four ranges of blends side by side
    from scarlet_amd import configure

    configure(hw_queues=int(os.environ.get("GPU_MAX_HW_QUEUES", "8")))
    if args.facade:
        return facade(args)
    if args.config == "cfg5":
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_cfg5

        return bench_cfg5.main(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    import torch
    from scarlet_amd import BlendBatch, dist as sdist

    # SCARLET_AMD_DIST_BACKEND=gloo + SCARLET_AMD_SHARE_GPU=1 let several ranks share one GPU
    # (exercises the multi-rank path on a single-GPU box; self_launch sets them)
    backend = os.environ.get("SCARLET_AMD_DIST_BACKEND")
    share = os.environ.get("SCARLET_AMD_SHARE_GPU") == "1"
    rank, local_rank, world = sdist.env_rank()
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if world > torch.cuda.device_count() and not share:
        # launched by torchrun on a box with fewer GPUs than ranks: same arrangement
        share, backend = True, "gloo"
    if share:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    sdist.init_process_group(backend=backend, device_index=local_rank)
    e_rel = 1e-3  # Blend.fit default; tolerance of the prox sub-iterations

    n_total = args.blends * world if args.weak else args.blends
    lo, hi = sdist.shard_range(n_total, rank, world)
    nb = hi - lo
    lite = args.loop != "blend"
    scenes = None
    if args.config == "cfg1":
        data, weights, comps, kernel = build_cfg1(nb)
    elif args.config == "cfg4":
        data, weights, comps, kernel = build_cfg4(nb)
    else:
        data, weights, comps, kernel, scenes = build_cfg3(lo, hi, local_rank,
                                                          args.loop if lite else None)
    K, Wm = args.steps, args.warmup
    batch = BlendBatch(
        data, weights, comps, kernel=None if args.null_renderer else kernel,
        max_iter=max(K, Wm) + (Wm + K if args.steady else 0) + 1, fft_shape=args.fft,
        device=local_rank, conv_path=args.conv_path,
        scheme="fista" if args.loop == "lite-fista" else "amsgrad", log_norm=not lite,
    )
    prox_max_iter = 1 if lite else 10  # lite applies the proximal operator once
    stream = torch.cuda.Stream(device=local_rank)
    batch.set_stream(stream.cuda_stream)
    batch.set_sub_ranges(args.sub_ranges)
    batch.save_state()  # parameters and optimizer state of iteration 0, kept on the device

    def fresh():
        """back to iteration 0 (device-to-device: the GPU does not idle before the timed region)"""
        batch.restore_state()

    def run(it0, n):
        batch.step(it0, n, e_rel=e_rel, prox_max_iter=prox_max_iter, check_convergence=False)

    def ramp(ms):
        """Keep the GPU busy with this batch's own iterations for `ms` milliseconds (state back
        to iteration 0 afterwards).  The driver's W = 5 warm-up iterations last 4.5 ms, the
        set-up before them (scene construction, uploads) and the oracle check before the
        roofline pass leave the GPU idle for seconds, and the power management needs ~20 ms of
        load to raise the clocks again: ten back-to-back windows of 20 iterations run at
        1 076 k, then 1 187 k blend-it/s each (tools/clock_ramp.py,
        profiles/r06_clock_ramp.txt).  What `value` should say is the rate of a fit, which
        lasts hundreds of iterations."""
        if ms <= 0:
            return
        t_end = time.perf_counter() + ms * 1e-3
        while time.perf_counter() < t_end:
            run(0, max(Wm, 5))
            torch.cuda.synchronize()
        fresh()

    # the clocks up (not a warm-up of anything the timed region reuses), then W untimed
    # warm-up iterations, then the state of iteration 0 again
    cold = None
    if args.ramp_ms > 0 and not args.steady:
        # the same protocol without the ramp first (rounds 1-5 measured this window): reported
        # beside `value` as config.cold_window_value, never as `value`
        run(0, Wm)
        torch.cuda.synchronize()
        fresh()
        torch.cuda.synchronize()
        sdist.barrier()
        t0 = time.perf_counter()
        run(0, K)
        torch.cuda.synchronize()
        sdist.barrier()
        cold = n_total * K / sdist.max_over_ranks(time.perf_counter() - t0)
        fresh()
    ramp(args.ramp_ms)
    run(0, Wm)
    torch.cuda.synchronize()
    it0 = 0
    if args.steady:
        it0 = Wm
    else:
        fresh()
    torch.cuda.synchronize()
    sdist.barrier()
    t0 = time.perf_counter()
    run(it0, K)
    torch.cuda.synchronize()
    sdist.barrier()
    elapsed = sdist.max_over_ranks(time.perf_counter() - t0)

    active, err = batch.status()
    assert err < 0, "non-finite parameters in blend %d" % err
    loss = batch.loss_history()
    if args.steady:
        loss = [l[Wm:] for l in loss]
    # the only collective: the packed per-blend records of all ranks
    rec = sdist.gather_records(sdist.pack_records(loss, batch.states(), K))
    comm = sdist.comm_info()  # backend / world size / devices of the live process group
    assert len(rec) == n_total and np.all(rec["n_iter"] == K)
    # parity at the benchmark's own size: first and last blend of every rank's shard against
    # the oracle, same K iterations (checker only, after the clock has stopped)
    parity = None
    if scenes is not None and not lite and not args.steady and not args.null_renderer:
        # (every rank enters the gather, a rank with an empty shard with an empty record)
        picks = sorted({0, nb - 1}) if nb else []
        early, late = oracle_check(scenes, picks, loss, K, e_rel) if picks else (0.0, 0.0)
        parity = sdist.gather_objects({"rank": rank, "blends": [lo + i for i in picks],
                                       "first_12": early, "all": late})
        if not any(p["blends"] for p in parity):
            parity = None

    # Roofline pass: the SAME K iterations again with HIP events around every kernel, all
    # of the rank's blends in ONE range.  In the timed region ranges of blends run on
    # streams of their own and their kernels share the chip, so the duration of a single
    # launch there says nothing about the kernel; `value` is not affected by this pass.
    ranges_timed = batch.sub_ranges()
    if not args.steady:
        ramp(args.ramp_ms)  # (the oracle check above left the GPU idle)
        fresh()
    batch.set_sub_ranges(1)
    batch.enable_timing(True)  # HIP events around every phase, on the batch stream
    run(it0 + (K if args.steady else 0), K)
    torch.cuda.synchronize()
    phases = {k: round(v, 4) for k, v in batch.timing().items()}  # mean ms over the K steps
    batch.enable_timing(False)
    fft_shape = batch.fft_shape
    conv_path = batch.conv_path  # "none" | "rocfft" | "fused": what the library runs

    if rank == 0:
        value = n_total * K / elapsed
        C, H, W = data.shape[1:]
        boxes = [c.morph.size for c in comps[0]]
        kb = 0 if args.null_renderer else kernel.shape[-3]
        Fy, Fx = fft_shape if not args.null_renderer else (0, 0)
        by = algorithmic_bytes(C, H, W, boxes, Fy, Fx, kb)
        by_survey = algorithmic_bytes(C, H, W, boxes, 180, 180, kb) if args.config == "cfg3" else by
        bytes_per = by["null"] if args.null_renderer else by["whole"]
        ms_iter = elapsed / K * 1e3
        fused = conv_path == "fused"
        k_name, k_bytes, k_ms = dominant_kernel(phases, conv_path, by)
        # `frac` / `achieved` price the bytes the dominant kernel must move at its fusion
        # boundary (<= 1 by construction); the SURVEY 8d byte model, whose transform passes
        # stay in LDS here, is reported beside it as frac_survey_model
        k_boundary = (by["conv_boundary"] if k_name == "fused_conv_kernel" else
                      by["update_boundary"] if k_name.startswith("update") else k_bytes)
        achieved = k_boundary * nb / (k_ms * 1e-3) / 1e9
        achieved_model = k_bytes * nb / (k_ms * 1e-3) / 1e9
        cnt = counters(k_name) if args.config == "cfg3" and nb == 1024 else {}
        # HBM bytes by the PMC counters, measured in this run where that is possible (N = 1,
        # the benchmark's own workload, the fused path); the committed summary otherwise
        live = None
        if (args.config == "cfg3" and world == 1 and conv_path == "fused" and not lite
                and not args.steady and not args.no_counters):
            live = live_hbm_counters(nb)
        live_name = "update_kernel_reg" if k_name.startswith("update") else k_name
        live_util = False
        if live and live_name in live:
            cnt = dict(cnt, **live[live_name])
            live_util = "valu_busy" in live[live_name]
            if live_util:
                # the rule of tools/hbm_counters.py, on this run's own counters
                k_ms_now = dominant_kernel(phases, conv_path, by)[2]
                hbm_now = live[live_name]["bytes_per_blend"] * nb / (k_ms_now * 1e-3) / 1e9 / HBM_PEAK_GBS
                cnt["bound"] = ("hbm" if hbm_now > 0.6 else
                                "valu-issue" if cnt.get("issue_busy", 0) > 0.85 else "latency-l2-lds")
        traffic = cnt["bytes_per_blend"] * nb if "bytes_per_blend" in cnt else None
        # Speed of light of one iteration of this rank's shard: what cannot be avoided is the
        # compulsory traffic B0 (data and weights once, parameters and moments once each way)
        # at 8 TB/s, and the butterflies of the four transforms at the f32 vector peak.
        counted = (args.config == "cfg3" and nb == 1024 and not args.null_renderer
                   and os.path.exists(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
        # every kernel of the iteration that the counter passes saw (no render_kernel since
        # the convolution kernel renders its own model rows)
        all_cnt = {k: counters(k) for k in ("fused_conv_kernel", "update_kernel_reg",
                                            "render_kernel") if counters(k)} if counted else {}
        if live:
            all_cnt = {k: dict(v) for k, v in live.items()
                       if k in ("fused_conv_kernel", "update_kernel_reg", "render_kernel")}
        hbm_ms = by["null"] * nb / (HBM_PEAK_GBS * 1e9) * 1e3
        flop_ms = (fft_flops(C, Fy, Fx) * nb / (F32_VECTOR_PEAK_TFLOPS * 1e12) * 1e3
                   if not args.null_renderer else 0.0)
        sol_ms = max(hbm_ms, flop_ms)
        measured_iter = (sum(c["bytes_per_blend"] for c in all_cnt.values())
                         if all_cnt and all("bytes_per_blend" in c for c in all_cnt.values()) else None)
        sol_frac = round(sol_ms / ms_iter, 4)
        hbm_whole = (round(measured_iter * nb / (ms_iter * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                     if measured_iter else None)
        frac = round(achieved / HBM_PEAK_GBS, 5)
        frac_model = round(achieved_model / HBM_PEAK_GBS, 5)
        roofline = {
            # the roofline `achieved` / `peak` are priced against (HBM bandwidth; no MFMA work
            # on this path), and what the PMC counters say limits the dominant kernel
            # ("unknown" without a counter summary for this workload)
            "bound": "hbm",
            "limiter": cnt.get("bound") or "unknown",
            # READ THESE FIRST.  frac_physical: share of the chip's physical limit the whole
            # iteration reaches (speed_of_light.sol_frac); hbm_frac_whole_iteration: HBM bytes
            # of the whole iteration by the PMC counters over ms_per_step against 8 TB/s (null
            # without a counter summary for this workload).  `frac` further down prices SURVEY
            # 8d's byte model for the dominant kernel and saturates (frac_note).
            "frac_physical": sol_frac,
            "hbm_frac_whole_iteration": hbm_whole,
            "speed_of_light": {
                "compulsory_bytes_per_blend_iteration": by["null"],
                "hbm_ms": round(hbm_ms, 4),
                "butterfly_flops_per_blend_iteration": (round(fft_flops(C, Fy, Fx))
                                                        if not args.null_renderer else 0),
                "valu_ms": round(flop_ms, 4),
                "sol_ms": round(sol_ms, 4),
                "sol_frac": sol_frac,
                "note": "sol_ms = max(compulsory bytes / 8 TB/s, butterfly flops / 157.3 TFLOP/s) "
                        "for this rank's %d blends; sol_frac = sol_ms / ms_per_step: the share of "
                        "the chip's physical limit the iteration reaches.  `frac` below prices the "
                        "SURVEY 8d byte model, most of whose bytes this design keeps in LDS." % nb,
            },
            "measured_hbm": {
                "bytes_per_blend_iteration": measured_iter,
                "over_compulsory": (round(measured_iter / by["null"], 3) if measured_iter else None),
                "frac_of_peak": hbm_whole,
                "per_kernel": {k: c.get("bytes_per_blend") for k, c in all_cnt.items()} or None,
                "source": ("this run: bench.py re-run under rocprofv3 --kernel-trace --pmc FETCH_SIZE / "
                           "WRITE_SIZE (separate passes, 6 iterations, one range of %d blends; KiB, "
                           "FETCH_SIZE doubled for gfx950)" % nb if live else
                           "profiles/hbm_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                           "passes of tools/collect_profiles.sh on 1024-blend launches "
                           "(committed constants, not measured in this run)" if all_cnt else None),
            },
            "counter_fields": {
                "fields": ([] if live_util else ["limiter", "valu_busy"] if live else
                           ["limiter", "traffic", "hbm_frac_measured", "valu_busy", "measured_hbm"]),
                "source": ("none: every field of this line is measured in this run (traffic, "
                           "hbm_frac_measured, measured_hbm, valu_busy, issue_busy and the limiter by "
                           "its own rocprofv3 --pmc passes)" if live_util else
                           "profiles/hbm_traffic.json (committed PMC summary); every other field is "
                           "measured in this run" + (" -- traffic, hbm_frac_measured and measured_hbm "
                                                     "by counter passes of this run" if live else "")),
            },
            "achieved": round(achieved, 2),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": frac,
            "frac_note": "bytes the dominant kernel must move through HBM at its fusion boundary "
                         "(data + weights + parameters in, gradient image out for the convolution; "
                         "parameters, moments and the gather of the gradient image for the update) "
                         "/ launch time / 8 TB/s; the SURVEY 8d byte model is frac_survey_model",
            # SURVEY 8d prices four FFT passes through HBM; the fused convolution keeps them
            # in LDS and moves ~1.1 MB of the 5.05 MB per blend, so this figure sits at or
            # above 1: it prices the survey's picture of the path, not the chip
            "frac_survey_model": frac_model,
            "achieved_survey_model": round(achieved_model, 2),
            "traffic": traffic,
            "kernel": k_name,
            "algorithmic_bytes_per_launch": k_boundary * nb,
            "algorithmic_bytes_per_blend": k_boundary,
            "survey_model_bytes_per_blend": k_bytes,
            "fft_shape_priced": [Fy, Fx],
            "ms_per_launch": round(k_ms, 4),
            "hbm_frac_measured": (round(traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
                                  if traffic else None),
            "valu_busy": cnt.get("valu_busy"),
            "issue_busy": cnt.get("issue_busy"),
            "flops_frac": (round(fft_flops(C, Fy, Fx) * nb / (k_ms * 1e-3) / 1e12
                                 / F32_VECTOR_PEAK_TFLOPS, 5)
                           if k_name == "fused_conv_kernel" else None),
            "measured": "HIP events on the batch stream, iterations %d..%d replayed in one "
                        "range of %d blends per launch (the timed region runs %d range(s) "
                        "concurrently)" % (it0 + (K if args.steady else 0),
                                           it0 + (K if args.steady else 0) + K - 1, nb,
                                           ranges_timed),
            "kernels": {
                "fused_conv_kernel": {
                    "ms": phases["conv"], "algorithmic_bytes_per_blend": by["conv"],
                    "frac": round(by["conv"] * nb / (phases["conv"] * 1e-3) / 8e12, 5)
                    if fused else None},
                "update_kernel_reg": {
                    "ms": phases["update"], "algorithmic_bytes_per_blend": by["update"],
                    "frac": round(by["update"] * nb / (phases["update"] * 1e-3) / 8e12, 5)},
            },
            "whole_iteration": {
                "algorithmic_bytes_per_blend_iteration": bytes_per,
                "ms": round(ms_iter, 4),
                "achieved": round(bytes_per * n_total / world / (ms_iter * 1e-3) / 1e9, 2),
                "frac": round(bytes_per * n_total / world / (ms_iter * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                "frac_survey_F180": round(by_survey["whole"] * n_total / world
                                          / (ms_iter * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
            },
            "phases_ms": phases,
        }
        if args.config == "cfg1":
            what = ("configs[0] scene (hsc_cosmos_35: 5x58x48, 10 components, boxes 21^2..61^2, "
                    "per-band 43x43 kernel) replicated: %d blends in total" % n_total)
        elif args.config == "cfg4":
            what = ("configs[3] scene (psf_unmatched_sim: 6x40x59, per-band 31x31 kernel, 3 "
                    "PointSources + 2 ExtendedSources in 71^2 / 81^2 boxes) replicated: %d blends "
                    "in total" % n_total)
        else:
            what = ("configs[2]: %d independent 5-band 128x128 blends in total%s, 10 "
                    "ExtendedSource components (41x41) each" % (
                        n_total, " (%d per GPU)" % args.blends if args.weak else ""))
        line = {
            "metric": "PGM iters/sec over batched blends; achieved HBM GB/s vs roofline",
            "value": round(value, 1),
            "unit": "blend-iterations/s",
            "n_gpus": world,
            "steps": K,
            "warmup": Wm,
            "ms_per_step": round(ms_iter, 4),
            "higher_is_better": True,
            "scaling": "weak" if args.weak else "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "%s, %s; iterations %d..%d of a fit from the initial parameters" % (
                    what, "NullRenderer ablation" if args.null_renderer
                    else "ConvolutionRenderer FFT %dx%d" % fft_shape, it0, it0 + K - 1),
                "loop": args.loop,
                "blends_total": n_total,
                "blends_per_gpu": [sdist.shard_range(n_total, r, world)[1]
                                   - sdist.shard_range(n_total, r, world)[0] for r in range(world)],
                "components_per_blend": len(comps[0]),
                "parallelism": "contiguous blend shards x%d, no data-path collective, one "
                               "all-gather of {n_iter, converged, logL, loss_hist[%d]} per "
                               "blend%s" % (world, K, " (ranks share GPUs over gloo: functional "
                                            "check, not a scaling point)" if share else ""),
                "sub_ranges_per_gpu": ranges_timed,
                "clock_ramp_ms": args.ramp_ms,
                "cold_window_value": None if cold is None else round(cold, 1),
                "clock_ramp_note": "untimed iterations of this batch before the W warm-up "
                                   "iterations and before the roofline pass, state restored: the "
                                   "GPU idles for seconds during set-up and needs ~20 ms of load "
                                   "to raise its clocks (profiles/r06_clock_ramp.txt: first "
                                   "20-iteration window 1 076 k, every later one 1 187 k "
                                   "blend-it/s); --ramp-ms 0 measures the cold window",
                "mean_logL": float(np.mean(rec["logL"])),
            },
            "roofline": roofline,
            "comm": comm,
            "parity": ({"checked_blends": [b for p in parity for b in p["blends"]],
                        "worst_rel_chi2_first_12_iterations": max(p["first_12"] for p in parity),
                        "worst_rel_chi2_all_iterations": max(p["all"] for p in parity),
                        "tolerance": [2e-5, 5e-4],
                        "against": "CPU oracle (oracle/pgm.py), same scenes, iterations 0..%d; "
                                   "checker only, after the timed region" % (K - 1)}
                       if parity else None),
        }
        if not args.no_cpu and not lite and world == 1:  # reported at N = 1 only
            line["cpu_baseline"] = cpu_baseline(args.config, args.cpu_blends,
                                                args.cpu_iters or K, e_rel)
            ref = os.path.join(ROOT, "profiles", "r03_reference_forward.json")
            if os.path.exists(ref):
                # SURVEY 8d(ii), measured once in the build container (the reference cannot
                # travel to the GPU box): the port's forward beside the reference's own
                with open(ref) as fh:
                    line["cpu_baseline"]["reference_forward_check"] = dict(
                        json.load(fh)["scenes"], source="profiles/r03_reference_forward.json "
                        "(oracle/refshim/time_reference_forward.py, build container, one thread)")
        print(json.dumps(line), flush=True)
    batch.close()
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
