"""the reference's own forward (Blend.get_model + Observation.render +
get_log_likelihood, under the container-only shims) timed beside the oracle's forward on
the same scenes, one thread, in the build container.  Shows that the CPU baseline
(`cpu_baseline.kind = "port"`) is not slower than the reference code it stands for.

    PYTHONPATH=/root/repo python oracle/refshim/time_reference_forward.py
"""
import os
import sys
import time

os.environ.setdefault("OMP_NUM_THREADS", "1")
import numpy as np  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from oracle import pgm  # noqa: E402
from oracle.refshim.load_reference import load  # noqa: E402

scarlet = load()
RESULTS = {}


def clock(fn, n):
    fn()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t) / n * 1e3


def reference_blend(images, weights, psfs, filters, comps):
    frame = scarlet.Frame(images.shape, psf=scarlet.GaussianPSF(sigma=(0.8,) * len(filters)),
                          channels=filters)
    obs = scarlet.Observation(images, psf=scarlet.ImagePSF(psfs), weights=weights,
                              channels=filters).match(frame)
    sources = []
    for sed, morph, (oy, ox) in comps:
        h, w = morph.shape
        box = scarlet.Box((len(filters), h, w), origin=(0, int(oy), int(ox)))
        spectrum = scarlet.TabulatedSpectrum(frame, sed.copy(), bbox=box[0])
        morphology = scarlet.ImageMorphology(frame, morph.copy(), bbox=box[1:])
        sources.append(scarlet.FactorizedComponent(frame, spectrum, morphology))
    return scarlet.Blend(sources, obs), obs


def report(name, images, weights, psfs, filters, comps, kernel):
    blend, obs = reference_blend(images, weights, psfs, filters, comps)

    def ref_forward():
        model = blend.get_model()
        return obs.get_log_likelihood(model)

    scene = pgm.Scene(images.shape, images, weights, kernel,
                      [pgm.Component(s.copy(), m.copy(), o) for s, m, o in comps])

    def oracle_forward():
        return scene.log_likelihood(scene.render(scene.get_model()))

    a, b = ref_forward(), oracle_forward()
    assert abs(a - b) < 1e-5 * abs(a), (a, b)
    t_ref, t_or = clock(ref_forward, 40), clock(oracle_forward, 40)
    print("%-28s reference forward %.2f ms, oracle forward %.2f ms (same logL %.3f)"
          % (name, t_ref, t_or, a))
    RESULTS[name] = {"reference_forward_ms": round(t_ref, 3), "oracle_forward_ms": round(t_or, 3),
                     "oracle_over_reference": round(t_or / t_ref, 3), "logL": a}


g = np.load(os.path.join(REPO, "tests", "golden", "hsc_cosmos_35.npz"))
comps = [(g["sed_%d" % k], g["morph_%d" % k], tuple(g["origin_%d" % k])) for k in range(int(g["n_comp"]))]
report("cfg 1 hsc_cosmos_35", g["images"], g["weights"], g["psfs"], list("grizy"), comps,
       g["diff_kernel"])

from scarlet_amd import synthetic  # noqa: E402

s = synthetic.make_blend(seed=1234)
comps = [(s["seds"][k], s["morphs"][k], tuple(s["origins"][k])) for k in range(len(s["seds"]))]
kern = synthetic.psfs()
report("cfg 2 synthetic 5x128x128", s["data"], s["weights"], s["obs_psf"], list("grizy"), comps, kern[2])

if "--json" in sys.argv:
    import json
    import platform

    cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][:1]
    out = {"what": "SURVEY 8d(ii): the reference's own forward (get_model + render + "
                   "get_log_likelihood under the container-only shims) beside the oracle's "
                   "forward, one thread, build container", "cpu": cpu, "python": platform.python_version(),
           "numpy": np.__version__, "scenes": RESULTS}
    with open(sys.argv[sys.argv.index("--json") + 1], "w") as fh:
        json.dump(out, fh, indent=1)
