"""SURVEY 8f-2 on the clock: `init_all_sources` per scene beside the fit it prepares.

For the quickstart scene (tests/golden/hsc_cosmos_35.npz) and for scenes of the benchmark
workload (configs[1]: 5 x 128 x 128, ten sources): wall time of the initialisation, its split
(source construction / set_spectra_to_match), the seam-1 calls it makes (count, bytes moved
host <-> device, wall time inside them) and the fit's wall time on the same scene.

    python tools/init_time.py [--scenes 4] [--out profiles/r06_init.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Seam1Counter:
    """Wraps the seam-1 entry points of the loaded library: calls, bytes, seconds."""

    NAMES = ("smi_prox_weighted_monotonic_f32", "smi_prox_weighted_monotonic_f64",
             "smi_apply_filter_f32", "smi_apply_filter_f64",
             "smi_prox_weighted_monotonic_many_f32", "smi_prox_weighted_monotonic_many_f64")

    def __init__(self):
        from scarlet_amd import _lib

        self.lib = _lib.load()
        self.calls = self.bytes = 0
        self.seconds = 0.0
        self.images = 0
        self._saved = {}
        for name in self.NAMES:
            if not hasattr(self.lib, name):
                continue
            fn = getattr(self.lib, name)
            self._saved[name] = fn

            def wrapped(*a, _fn=fn, _name=name):
                t0 = time.perf_counter()
                rc = _fn(*a)
                self.seconds += time.perf_counter() - t0
                self.calls += 1
                if "many" in _name:
                    n_img, n_pix = int(a[0]), int(a[2])
                    width = 8 if _name.endswith("f64") else 4
                    self.images += n_img
                    # images both ways + weights (8 x n_pix) + order per image
                    self.bytes += n_img * n_pix * (2 * width + 8 * width + 4)
                elif "monotonic" in _name:
                    n_pix = int(a[6])
                    width = 8 if _name.endswith("f64") else 4
                    self.images += 1
                    self.bytes += n_pix * (2 * width + 8 * width + 4)
                return rc

            setattr(self.lib, name, wrapped)

    def reset(self):
        self.calls = self.bytes = self.images = 0
        self.seconds = 0.0

    def snapshot(self):
        return dict(seam1_calls=self.calls, seam1_images=self.images, seam1_bytes=self.bytes,
                    seam1_ms=round(self.seconds * 1e3, 3))


def quickstart():
    import scarlet_amd as scarlet

    hsc = np.load(os.path.join(ROOT, "tests", "golden", "hsc_cosmos_35.npz"), allow_pickle=True)
    filters = list("grizy")
    frame = scarlet.Frame(hsc["images"].shape, psf=scarlet.GaussianPSF(sigma=(0.8,) * 5), channels=filters)
    obs = scarlet.Observation(hsc["images"], psf=scarlet.ImagePSF(hsc["psfs"].copy()),
                              weights=hsc["weights"], channels=filters).match(frame)
    return frame, obs, [tuple(c) for c in hsc["centers"]], dict(max_components=2, min_snr=50, thresh=1)


def synthetic_scene(seed):
    import scarlet_amd as scarlet
    from scarlet_amd import synthetic

    kern = synthetic.psfs()
    s = synthetic.make_blend(seed, kernel=kern)
    channels = list("grizy")
    frame = scarlet.Frame((5, synthetic.H, synthetic.W),
                          psf=scarlet.GaussianPSF(sigma=(synthetic.SIGMA_MODEL,) * 5), channels=channels)
    obs = scarlet.Observation(s["data"], psf=scarlet.ImagePSF(np.repeat(kern[0], 5, axis=0)),
                              weights=s["weights"], channels=channels).match(frame)
    centers = [(float(o[0] + m.shape[0] // 2), float(o[1] + m.shape[1] // 2))
               for o, m in zip(s["origins"], s["morphs"])]
    return frame, obs, centers, dict(max_components=1, min_snr=50, thresh=1)


def measure(make, counter, repeats=3, fit_iters=100):
    import scarlet_amd as scarlet
    from scarlet_amd import initialization as init

    best = None
    for _ in range(repeats):
        frame, obs, centers, kw = make()
        counter.reset()
        t0 = time.perf_counter()
        sources, skipped = init.init_all_sources(frame, centers, obs, fallback=True, silent=True,
                                                 set_spectra=False, **kw)
        t1 = time.perf_counter()
        construct = counter.snapshot()
        init.set_spectra_to_match(sources, obs)
        t2 = time.perf_counter()
        blend = scarlet.Blend(sources, obs)
        n_it, logL = blend.fit(fit_iters, e_rel=1e-4)
        t3 = time.perf_counter()
        row = dict(n_sources=len(sources), n_skipped=len(skipped),
                   frame=list(frame.shape),
                   init_ms=round((t2 - t0) * 1e3, 3),
                   sources_ms=round((t1 - t0) * 1e3, 3),
                   set_spectra_ms=round((t2 - t1) * 1e3, 3),
                   fit_ms=round((t3 - t2) * 1e3, 3), fit_iterations=int(n_it),
                   **construct)
        if best is None or row["init_ms"] < best["init_ms"]:
            best = row
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=4)
    ap.add_argument("--out", default=None)
    ap.add_argument("--profile", action="store_true", help="cProfile of one cfg-2 scene's initialisation")
    args = ap.parse_args()
    import scarlet_amd as scarlet  # noqa: F401

    counter = Seam1Counter()
    measure(quickstart, counter, repeats=1, fit_iters=5)  # library load, plan caches, first launches
    if args.profile:
        import cProfile
        import pstats
        from scarlet_amd import initialization as init

        measure(lambda: synthetic_scene(1234), counter, repeats=1, fit_iters=2)
        frame, obs, centers, kw = synthetic_scene(1235)
        prof = cProfile.Profile()
        prof.enable()
        init.init_all_sources(frame, centers, obs, fallback=True, silent=True, set_spectra=True, **kw)
        prof.disable()
        pstats.Stats(prof).sort_stats("cumulative").print_stats(40)
        return
    out = dict(what="init_all_sources (SURVEY 8f-2) beside Blend.fit(100, e_rel=1e-4) per scene; best "
                    "of 3; seam1_* = calls of the monotonic sweep through the C ABI during source "
                    "construction (count, images swept, bytes moved host<->device incl. tables, wall ms "
                    "inside the calls)",
               quickstart=measure(quickstart, counter))
    rows = [measure(lambda s=1234 + i: synthetic_scene(s), counter) for i in range(args.scenes)]
    out["configs1_scenes"] = rows
    out["configs1_mean"] = {k: round(float(np.mean([r[k] for r in rows])), 3)
                            for k in ("init_ms", "sources_ms", "set_spectra_ms", "fit_ms", "seam1_ms")}
    text = json.dumps(out, indent=1)
    print(text)
    if args.out:
        with open(args.out, "w") as fh:
            fh.write(text + "\n")


if __name__ == "__main__":
    main()
