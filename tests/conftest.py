import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def hsc():
    return golden("hsc_cosmos_35")


def hsc_scene(g, dtype64=False, state_dtype=np.float64):
    """oracle.pgm.Scene of the quickstart blend from the golden fixture.
    ``state_dtype=np.float32``: the oracle's float32-state mode (device arithmetic)."""
    from oracle import pgm

    n = int(g["n_comp"])
    comps = []
    for k in range(n):
        sed = g["sed64_%d" % k] if dtype64 else g["sed_%d" % k]
        morph = g["morph64_%d" % k] if dtype64 else g["morph_%d" % k]
        comps.append(
            pgm.Component(sed.copy(), morph.copy(), g["origin_%d" % k],
                          sed_min_step=g["min_step_%d" % k], source=int(g["source_of"][k]),
                          state_dtype=state_dtype)
        )
    dt = np.float64 if dtype64 else np.float32
    images = g["images"].astype(dt)
    weights = g["weights"].astype(dt)
    kernel = g["diff_kernel"].astype(dt)
    return pgm.Scene(images.shape, images, weights, kernel, comps, dtype=dt)


def point_scene(g, dtype64=False):
    """oracle.pgm.Scene of the point-source tutorial blend (stars = PointComponent)."""
    from oracle import pgm

    tag = "f64_" if dtype64 else ""
    # model PSF: GaussianPSF(0.9), or MoffatPSF(alpha, beta, boxsize=15) (point_source_moffat)
    psf = dict(sigma=0.9) if "moffat" not in g else dict(
        sigma=float(g["moffat"][0]), beta=float(g["moffat"][1]), boxsize=15)
    if "psf_image" in g:  # ImagePSF
        psf = dict(sigma=0.0, image=g["psf_image"])
    comps = []
    for k in range(int(g["n_src"])):
        sed = g["%ssed_%d" % (tag, k)].copy()
        if g["is_star"][k]:
            comps.append(pgm.PointComponent(sed, g["%scenter_%d" % (tag, k)],
                                            sed_min_step=g["min_step_%d" % k], **psf))
        else:
            comps.append(pgm.Component(sed, g["%smorph_%d" % (tag, k)].copy(),
                                       g["%sorigin_%d" % (tag, k)],
                                       sed_min_step=g["min_step_%d" % k]))
    dt = np.float64 if dtype64 else np.float32
    images = g["images"].astype(dt)
    weights = np.full(images.shape, 0.25, dtype=dt)
    return pgm.Scene(images.shape, images, weights, g["diff_kernel"].astype(dt), comps, dtype=dt)


def shifting_scene(g, hsc, dtype64=False):
    """oracle.pgm.Scene of the quickstart blend with free Fourier shifts
    (ExtendedSource(shifting=True)); observation from the hsc_cosmos_35 fixture."""
    from oracle import pgm

    comps = []
    for k in range(int(g["n_comp"])):
        sed = g["sed64_%d" % k] if dtype64 else g["sed_%d" % k]
        morph = g["morph64_%d" % k] if dtype64 else g["morph_%d" % k]
        comps.append(pgm.Component(sed.copy(), morph.copy(), g["origin_%d" % k],
                                   sed_min_step=g["min_step_%d" % k],
                                   source=int(g["source_of"][k]), shift=g["shift_%d" % k]))
    dt = np.float64 if dtype64 else np.float32
    images = hsc["images"].astype(dt)
    return pgm.Scene(images.shape, images, hsc["weights"].astype(dt),
                     hsc["diff_kernel"].astype(dt), comps, dtype=dt)


def lite_scene(g, hsc, kind):
    """oracle.lite.LiteScene of the quickstart blend, components as the lite goldens
    were started (hsc_cosmos_35 initial sources, bg_thresh 0.25)."""
    from oracle import lite

    images = hsc["images"].astype(np.float32)
    weights = hsc["weights"].astype(np.float32)
    comps = []
    for k in range(int(g["n_comp"])):
        sed = hsc["sed_%d" % k].astype(np.float32).copy()
        morph = hsc["morph_%d" % k].astype(np.float32).copy()
        kw = dict(fista_step=float(g["fista_step"][k])) if kind == "fista" else dict(
            sed_min_step=g["noise_rms"] / 10)
        comps.append(lite.LiteComponent(sed, morph, hsc["origin_%d" % k], g["noise_rms"],
                                        bg_thresh=0.25, kind=kind, **kw))
    return lite.LiteScene(images, weights, g["diff_kernel"], comps)
