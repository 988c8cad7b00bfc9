"""Parity of the HIP path (through the C ABI) with the CPU oracle, on a real
MI355X.  Tolerance: 1e-5 relative (float32), as BASELINE.json's north_star
states; integer/index work and the monotonic sweep / apply_filter are bit exact.
"""

import ctypes
import os

import numpy as np
import pytest
from numpy.testing import assert_allclose, assert_array_equal

from conftest import golden, hsc_scene

pytestmark = pytest.mark.gpu

RTOL = 1e-5


def rel_err(a, b):
    return np.abs(np.asarray(a, dtype=np.float64) - b).max() / max(np.abs(b).max(), 1e-300)


def assert_loss_close(loss, ref_loss, log_norm, rtol=RTOL):
    """loss = log_norm + chi2/2 can pass through zero: compare the chi2 part"""
    a = np.asarray(loss, dtype=np.float64) - log_norm
    b = np.asarray(ref_loss, dtype=np.float64) - log_norm
    assert a.shape == b.shape
    assert np.all(np.abs(a - b) <= rtol * np.abs(b)), (a, b)


@pytest.fixture(scope="module")
def amd():
    import scarlet_amd
    from scarlet_amd import _lib

    _lib.load()
    assert _lib.load().smi_device_count() >= 1
    return scarlet_amd


# ---------------------------------------------------------------- seam 1
MODES = [("flat", 0.1), ("angle", 0.0), ("nearest", 0.0), ("angle", 0.25)]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(5, 5), (21, 21), (41, 41), (31, 41), (22, 30), (81, 81)])
def test_sweep_bit_exact(amd, dtype, shape):
    from oracle import proxops

    rng = np.random.default_rng(shape[0] * 100 + shape[1])
    for mode, g in MODES:
        w, didx, off = proxops.monotonic_operator(shape, mode, (shape[0] // 2, shape[1] // 2))
        x0 = rng.random(shape).astype(dtype)
        want = proxops.sweep(x0.copy(), w, off, didx, g)
        got = amd.operator._native_sweep(x0.copy(), w, off, didx, g)
        assert_array_equal(got, want)


@pytest.mark.parametrize("dtype,shape,center", [(np.float64, (190, 170), (70, 101)),
                                                (np.float32, (282, 282), (116, 162))])
def test_sweep_bit_exact_beyond_the_lds(amd, dtype, shape, center):
    """Initialisation sweeps the whole detection image about the source's pixel (not the
    middle): images larger than the 160 KiB LDS take the global-memory kernel."""
    from oracle import proxops

    assert shape[0] * shape[1] * np.dtype(dtype).itemsize > 160 * 1024
    rng = np.random.default_rng(shape[0])
    for mode, g in (("angle", 0.0), ("flat", 0.1)):
        w, didx, off = proxops.monotonic_operator(shape, mode, center)
        x0 = rng.random(shape).astype(dtype)
        want = proxops.sweep(x0.copy(), w, off, didx, g)
        got = amd.operator._native_sweep(x0.copy(), w, off, didx, g)
        assert_array_equal(got, want)


@pytest.mark.parametrize("dtype,shape", [(np.float64, (58, 48)), (np.float32, (128, 128)),
                                         (np.float64, (128, 128)), (np.float32, (7, 9)),
                                         (np.float64, (150, 150))])
def test_many_sweeps_in_one_launch_equal_the_single_sweeps(amd, dtype, shape):
    """``smi_prox_weighted_monotonic_many_*``: the detection images of a scene's sources, each
    about a centre of its own (on the edge and in a corner too), made monotonic in one launch --
    bit for bit what the one-image entry point and the oracle give image by image.  150 x 150
    doubles are beyond the LDS: that call falls back to image-by-image launches."""
    from oracle import proxops

    rng = np.random.default_rng(shape[0] * 7 + shape[1])
    n = 6
    centers = [(int(rng.integers(0, shape[0])), int(rng.integers(0, shape[1]))) for _ in range(n - 2)]
    centers += [(0, 0), (shape[0] - 1, shape[1] // 2)]
    for mode, g in (("flat", 0.0), ("angle", 0.1)):
        x0 = rng.random((n,) + shape).astype(dtype)
        got = amd.operator.prox_weighted_monotonic_many(x0.copy(), centers, mode, g)
        for i, c in enumerate(centers):
            w, didx, off = proxops.monotonic_operator(shape, mode, c)
            assert_array_equal(got[i], proxops.sweep(x0[i].copy(), w, off, didx, g))
            one = amd.operator.prox_weighted_monotonic(shape, mode, g, center=c)
            assert_array_equal(got[i], one(x0[i].copy(), 0))
    assert amd.operator.prox_weighted_monotonic_many(x0[:0].copy(), [], "flat", 0).shape[0] == 0


def test_sweep_reference_known_answers(amd):
    """reference tests/test_constraint.py:92-135 through the product classes"""
    from test_oracle_golden import MONO_NEAREST, MONO_ANGLE, MONO_ANGLE_G25

    X = np.arange(25, dtype=float).reshape(5, 5)
    for mode, g, truth in (("nearest", 0, MONO_NEAREST), ("angle", 0, MONO_ANGLE),
                           ("angle", 0.25, MONO_ANGLE_G25)):
        c = amd.MonotonicityConstraint(neighbor_weight=mode, min_gradient=g)
        np.testing.assert_almost_equal(c(X.copy(), 0), truth)


def test_sweep_golden_from_reference(amd):
    g = golden("operator_tables")
    for tag in ("21x21", "31x41"):
        x0 = g["sweep_in_" + tag]
        for mode, gr in MODES:
            c = amd.MonotonicityConstraint(neighbor_weight=mode, min_gradient=gr)
            assert_array_equal(c(x0.copy(), 0), g["sweep_{}_{}_{}".format(mode, gr, tag)])


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_sweep_and_mask_operators_independent_known_answers(amd, dtype):
    """seam 1 against answers that do not pass through the oracle's C files: the scalar
    Python transcription of the reference's sweep loop on the reference's own set-up
    tables, and the hand-derived known answers of the two mask operators
    (tests/mask_kats.py)"""
    import mask_kats
    from scarlet_amd import operator

    g = golden("operator_tables")
    for tag in ("7x9", "21x21"):
        h, w = map(int, tag.split("x"))
        off = np.array([-w - 1, -w, -w + 1, -1, 1, w - 1, w, w + 1], dtype=np.int32)
        rng = np.random.default_rng(h * 100 + w)
        didx = g["didx_" + tag][1:]
        for mode, gr in MODES:
            wts = g["w_{}_{}".format(mode, tag)]
            x0 = rng.random((h, w)).astype(dtype)
            want = mask_kats.sweep_transcription(x0.copy(), wts, off, didx, gr)
            got = operator._native_sweep(x0.copy(), wts, off, didx, gr)
            assert_array_equal(got, want)
    for name, img, (i, j), var, thr, unc_w, orp_w, b_w in mask_kats.valid_pixel_cases(dtype):
        unchecked = np.ones(img.shape, dtype=bool)
        unchecked[i, j] = False
        orphans = np.zeros(img.shape, dtype=bool)
        bounds = np.array([i, i, j, j], dtype=np.int32)
        operator.get_valid_monotonic_pixels(i, j, img, unchecked, orphans, var, bounds, thr)
        assert_array_equal(unchecked, unc_w, err_msg=name)
        assert_array_equal(orphans, orp_w, err_msg=name)
        assert bounds.tolist() == b_w, name
    for name, model, unc, orp, oi, oj, rec, b0, m_w, unc_w, orp_w, b_w in \
            mask_kats.interpolation_cases(dtype):
        model, unc, orp = model.copy(), unc.copy(), orp.copy()
        bounds = np.array(b0, dtype=np.int32)
        operator.linear_interpolate_invalid_pixels(oi, oj, unc, model, orp, 0.0, rec, bounds)
        assert_array_equal(model, m_w, err_msg=name)
        assert_array_equal(unc, unc_w, err_msg=name)
        assert_array_equal(orp, orp_w, err_msg=name)
        assert bounds.tolist() == b_w, name


def test_sweep_empty_and_degenerate(amd):
    from oracle import proxops

    # 1x1 image: empty sweep order
    x = np.array([[3.0]])
    w, didx, off = proxops.monotonic_operator((1, 1), "angle", (0, 0))
    assert didx.size == 0
    assert_array_equal(amd.operator._native_sweep(x.copy(), w, off, didx, 0.0), x)
    # a single row
    w, didx, off = proxops.monotonic_operator((1, 9), "flat", (0, 4))
    x = np.arange(9, dtype=np.float64)[None]
    want = proxops.sweep(x.copy(), w, off, didx, 0.1)
    assert_array_equal(amd.operator._native_sweep(x.copy(), w, off, didx, 0.1), want)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_apply_filter_bit_exact(amd, dtype):
    from oracle import fftconv, proxops
    from scarlet_amd import _lib

    rng = np.random.default_rng(1)
    img = rng.standard_normal((40, 59)).astype(dtype)
    ker = rng.standard_normal((7, 5)).astype(dtype)
    ys, ye, xs, xe = fftconv.filter_bounds(ker)
    vals = np.ascontiguousarray(ker.reshape(-1))
    want = np.empty_like(img)
    lib = proxops._lib()
    ofn = lib.oracle_apply_filter_f32 if dtype == np.float32 else lib.oracle_apply_filter_f64
    ofn.restype = None
    vp = ctypes.c_void_p
    ofn(img.ctypes.data_as(vp), 40, 59, vals.ctypes.data_as(vp), vals.size,
        ys.ctypes.data_as(vp), ye.ctypes.data_as(vp), xs.ctypes.data_as(vp),
        xe.ctypes.data_as(vp), want.ctypes.data_as(vp))
    got = np.empty_like(img)
    ct = ctypes.c_float if dtype == np.float32 else ctypes.c_double
    fn = _lib.load().smi_apply_filter_f32 if dtype == np.float32 else _lib.load().smi_apply_filter_f64
    _lib.check(fn(_lib.ptr(img, ct), 40, 59, _lib.ptr(vals, ct), vals.size,
                  _lib.ptr(ys, ctypes.c_int32), _lib.ptr(ye, ctypes.c_int32),
                  _lib.ptr(xs, ctypes.c_int32), _lib.ptr(xe, ctypes.c_int32), _lib.ptr(got, ct)))
    assert_array_equal(got, want)
    assert_allclose(got, fftconv.apply_filter(img, ker), rtol=1e-4, atol=1e-5)


def _bumpy(rng, shape, dtype):
    yy, xx = np.mgrid[: shape[0], : shape[1]]
    cy, cx = shape[0] // 2, shape[1] // 2
    img = np.exp(-0.5 * (((yy - cy) / (0.2 * shape[0])) ** 2 + ((xx - cx) / (0.22 * shape[1])) ** 2))
    img += 0.5 * np.exp(-0.5 * (((yy - cy // 2) / 2.5) ** 2 + ((xx - 1.6 * cx) / 3.0) ** 2))
    return (img + rng.normal(0, 0.03, shape)).astype(dtype)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(45, 39), (58, 48), (7, 9), (1, 12), (101, 120)])
def test_mask_operators_bit_exact(amd, dtype, shape):
    """get_valid_monotonic_pixels / linear_interpolate_invalid_pixels through the C ABI
    against the C restatement of operators_pybind11.cc:61-232: identical unchecked /
    orphans maps, bounds and interpolated values (the parallel relaxation on the GPU
    reaches the same fixed point as the depth-first recursion)"""
    from oracle import proxops
    from scarlet_amd import operator

    rng = np.random.default_rng(shape[0] * 100 + shape[1])
    img = _bumpy(rng, shape, dtype)
    starts = [(shape[0] // 2, shape[1] // 2), (0, 0), (shape[0] - 1, shape[1] // 3)]
    for (i, j), variance, thresh in zip(starts, (0.0, 0.02, 0.0), (0.0, 0.0, 0.05)):
        state = []
        for mod in (proxops, operator):
            unchecked = np.ones(shape, dtype=bool)
            unchecked[i, j] = False
            if shape[0] > 5:
                unchecked[2, :3] = False  # pixels some earlier pass already settled
            orphans = np.zeros(shape, dtype=bool)
            bounds = np.array([i, i, j, j], dtype=np.int32)
            mod.get_valid_monotonic_pixels(i, j, img, unchecked, orphans, variance, bounds, thresh)
            model = img.copy()
            first = (unchecked.copy(), orphans.copy(), bounds.copy())
            for recursive in (True, False):
                oi, oj = np.where(orphans)
                mod.linear_interpolate_invalid_pixels(oi, oj, unchecked, model, orphans, variance,
                                                      recursive, bounds)
            state.append(first + (unchecked, orphans, bounds, model))
        for a, b in zip(*state):
            assert_array_equal(a, b)
        assert state[0][0].sum() < img.size  # something was reached


def test_prox_monotonic_mask_and_use_mask_constraint(amd):
    """operator.prox_monotonic_mask (operator.py:131-176) and
    MonotonicityConstraint(use_mask=True) (constraint.py:225-232) on the GPU operators
    against the oracle's restatement"""
    from oracle import proxops
    from scarlet_amd import operator
    import scarlet_amd as scarlet

    rng = np.random.default_rng(5)
    for dtype in (np.float32, np.float64):
        img = _bumpy(rng, (51, 47), dtype)
        for max_iter, radius, variance in ((0, 1, 0.0), (3, 1, 0.0), (2, 0, 0.01)):
            got = operator.prox_monotonic_mask(img.copy(), 0, (25, 23), center_radius=radius,
                                               variance=variance, max_iter=max_iter)
            want = proxops.prox_monotonic_mask(img.copy(), 0, (25, 23), center_radius=radius,
                                               variance=variance, max_iter=max_iter)
            for a, b in zip(got, want):
                assert_array_equal(a, b)
        morph = img.copy()
        out = scarlet.MonotonicityConstraint("angle", 0, use_mask=True)(morph, 0)
        ref = img.copy()
        w, didx, off = proxops.monotonic_operator(ref.shape, "angle", (25, 23))
        proxops.sweep(ref, w, off, didx, 0)
        valid, masked, _ = proxops.prox_monotonic_mask(img.copy(), 0, (25, 23), center_radius=0,
                                                       variance=0, max_iter=0)
        ref[valid] = masked[valid]
        assert_array_equal(out, ref)


# ---------------------------------------------------------------- seam 2
PATHS = ["rocfft", "fused"]


def hsc_batch(amd, g, **kw):
    n = int(g["n_comp"])
    comps = [
        amd.ComponentSpec(g["sed_%d" % k], g["morph_%d" % k], g["origin_%d" % k],
                          sed_min_step=g["min_step_%d" % k])
        for k in range(n)
    ]
    return amd.BlendBatch(g["images"][None], g["weights"][None], [comps],
                          kernel=g["diff_kernel"], **kw)


@pytest.mark.parametrize("path", PATHS)
def test_hsc_forward_vs_golden_and_oracle(amd, hsc, path):
    batch = hsc_batch(amd, hsc, conv_path=path)
    # rocFFT path: the reference's shape (fft.py:116-167 on (58,48)+(43,43)+3);
    # fused path: the smallest alias-free supported shape >= N + P//2
    assert batch.fft_shape == ((108, 96) if path == "rocfft" else (80, 80))
    model, rendered, logL = batch.forward()
    assert rel_err(model[0], hsc["model"]) < RTOL
    assert rel_err(rendered[0], hsc["rendered"]) < RTOL
    chi2_ref = -(float(hsc["logL"]) + float(hsc["log_norm"]))
    sc = hsc_scene(hsc)
    chi2 = -(logL[0] + sc.log_norm)
    assert abs(chi2 - chi2_ref) < RTOL * abs(chi2_ref)
    assert abs(logL[0] - float(hsc["logL"])) < 1e-5 * abs(float(hsc["logL"]))


def grad_scales(sc):
    """sum |G||morph| and sum |sed||G| per component: the magnitudes the float32
    sums are taken over (the gradients themselves cancel to much less)"""
    model = sc.get_model()
    G = np.abs(sc.model_gradient(sc.render(model)))
    out = []
    for c in sc.components:
        h, w = c.morph.shape
        boxed = np.zeros((sc.frame_shape[0], h, w))
        fs, bs = sc.box_slices(c)
        boxed[bs] = G[fs]
        out.append((np.einsum("cyx,yx->c", boxed, np.abs(c.morph)).max(),
                    np.einsum("c,cyx->yx", np.abs(c.sed), boxed).max()))
    return out


@pytest.mark.parametrize("path", PATHS)
def test_hsc_gradient_vs_oracle(amd, hsc, path):
    batch = hsc_batch(amd, hsc, conv_path=path)
    g_sed, g_morph = batch.gradient()
    sc = hsc_scene(hsc)
    _, grads = sc.loss_and_gradients()
    for k, ((gs, gm), (s_sed, s_morph)) in enumerate(zip(grads, grad_scales(sc))):
        assert np.abs(g_sed[k] - gs).max() < RTOL * s_sed, k
        assert np.abs(g_morph[k] - gm).max() < RTOL * s_morph, k


@pytest.mark.parametrize("path", PATHS)
def test_hsc_steps_vs_oracle(amd, hsc, path):
    n_it = 5
    batch = hsc_batch(amd, hsc, max_iter=16, conv_path=path)
    batch.step(0, n_it, e_rel=1e-3)
    sed, morphs = batch.parameters()
    mom = batch.moments()
    sc = hsc_scene(hsc)
    for it in range(n_it):
        sc.step(it, 1e-3)
    loss = batch.loss_history()[0]
    assert_loss_close(loss, sc.loss, sc.log_norm)
    for k, c in enumerate(sc.components):
        assert rel_err(sed[k], c.sed) < 1e-4, k
        assert np.abs(morphs[k] - c.morph).max() < 1e-4, k
        assert rel_err(mom["m_sed"][k], c.m_sed) < 1e-3
        assert rel_err(mom["v_morph"][k], c.v_morph) < 1e-3


@pytest.mark.parametrize("path", PATHS)
def test_first_step_exact_structure(amd, hsc, path):
    """one iteration from identical state: every parameter within 1e-5"""
    batch = hsc_batch(amd, hsc, max_iter=4, conv_path=path)
    batch.step(0, 1, e_rel=1e-3)
    sed, morphs = batch.parameters()
    sc = hsc_scene(hsc)
    sc.step(0, 1e-3)
    for k, c in enumerate(sc.components):
        assert rel_err(sed[k], c.sed) < RTOL, k
        assert np.abs(morphs[k] - c.morph).max() < RTOL, k
        assert morphs[k].max() == 1.0 and morphs[k].min() >= 0.0


@pytest.mark.parametrize("path", PATHS)
def test_hsc_fit_follows_the_oracle_for_all_iterations(amd, hsc, path):
    """Whole fit of the quickstart scene (BASELINE configs[0]: 100 iterations max,
    e_rel 1e-4) against the reference-faithful oracle (float32 parameters, float64
    optimizer state, blend.py:155-160) and against its float32-state mode, which restates
    the device's precision.  Measured on MI355X (tools/fit_parity.py): identical iteration
    count (76); relative chi^2 difference <= 8e-6 over the first 12 iterations, 1.2e-7 /
    8e-7 (fused / rocFFT) at the end, and a transient of 1.4e-4 / 2.1e-4 around iteration
    25.  The transient is not the float32 state: the two oracle modes differ from EACH
    OTHER by 2.7e-4 there and by 1.2e-7 at the end (tests/test_oracle_golden.py::
    test_float32_state_mode_of_the_oracle), with identical sub-iteration counts -- the scene
    passes through a few ill-conditioned iterations in which any 1e-7 perturbation grows a
    thousandfold and then dies out.  north_star's 1e-5 holds for the result of the fit."""
    batch = hsc_batch(amd, hsc, max_iter=100, conv_path=path)
    n_iter, logL = batch.fit(max_iter=100, e_rel=1e-4)
    loss = batch.loss_history()[0]
    assert len(loss) == n_iter[0] and -loss[-1] == logL[0]
    for state_dtype, whole, final in ((np.float64, 5e-4, RTOL), (np.float32, 5e-4, 2e-6)):
        sc = hsc_scene(hsc, state_dtype=state_dtype)
        n_ref, logL_ref = sc.fit(max_iter=100, e_rel=1e-4)
        assert int(n_iter[0]) == n_ref  # the stopping rule fires in the same iteration
        chi, ref = loss - sc.log_norm, np.array(sc.loss) - sc.log_norm
        rel = np.abs(chi - ref) / np.abs(ref)
        assert rel[:12].max() < 2e-5
        assert rel.max() < whole
        assert rel[-1] < final
    assert -loss[-1] > -loss[0]


@pytest.mark.parametrize("path", PATHS)
def test_synthetic_batch_vs_oracle(amd, path):
    from oracle import pgm
    from scarlet_amd import synthetic

    kern = synthetic.psfs()
    scenes = [synthetic.make_blend(1234 + b, kernel=kern) for b in range(3)]
    comps = [
        [amd.ComponentSpec(s["seds"][k], s["morphs"][k], s["origins"][k],
                           sed_min_step=s["noise_rms"]) for k in range(10)]
        for s in scenes
    ]
    data = np.stack([s["data"] for s in scenes])
    weights = np.stack([s["weights"] for s in scenes])
    batch = amd.BlendBatch(data, weights, comps, kernel=kern[2], max_iter=8, conv_path=path)
    assert batch.fft_shape == ((180, 180) if path == "rocfft" else (160, 160))
    g = golden("synthetic_cfg2")
    model, rendered, logL = batch.forward()
    assert rel_err(model[0], g["model"]) < RTOL
    assert rel_err(rendered[0], g["rendered"]) < RTOL
    batch.step(0, 3, e_rel=1e-3)
    sed, morphs = batch.parameters()
    losses = batch.loss_history()
    for b, s in enumerate(scenes):
        sc = pgm.Scene(
            s["data"].shape, s["data"], s["weights"], s["diff_kernel"],
            [pgm.Component(s["seds"][k].copy(), s["morphs"][k].copy(), s["origins"][k],
                           sed_min_step=s["noise_rms"]) for k in range(10)],
        )
        for it in range(3):
            sc.step(it, 1e-3)
        assert_loss_close(losses[b], sc.loss, sc.log_norm)
        for k, c in enumerate(sc.components):
            assert rel_err(sed[b * 10 + k], c.sed) < 1e-4
            assert np.abs(morphs[b * 10 + k] - c.morph).max() < 1e-4


def test_null_renderer_vs_oracle(amd):
    from oracle import pgm
    from scarlet_amd import synthetic

    s = synthetic.make_blend(99)
    comps = [amd.ComponentSpec(s["seds"][k], s["morphs"][k], s["origins"][k],
                               sed_min_step=s["noise_rms"]) for k in range(10)]
    batch = amd.BlendBatch(s["data"][None], s["weights"][None], [comps], kernel=None, max_iter=8)
    sc = pgm.Scene(
        s["data"].shape, s["data"], s["weights"], None,
        [pgm.Component(s["seds"][k].copy(), s["morphs"][k].copy(), s["origins"][k],
                       sed_min_step=s["noise_rms"]) for k in range(10)],
    )
    model, rendered, logL = batch.forward()
    assert_array_equal(model, rendered)
    assert rel_err(model[0], sc.get_model()) < RTOL
    batch.step(0, 2, e_rel=0.0)  # tolerance 0: always prox_max_iter sub-iterations
    for it in range(2):
        sc.step(it, 0.0)
    assert_loss_close(batch.loss_history()[0], sc.loss, sc.log_norm)


@pytest.mark.parametrize("path", PATHS)
def test_per_band_kernel_and_ragged_boxes(amd, path):
    """cfg 4 shapes: (6,40,59) frame, (6,31,31) kernel, boxes overhanging the frame"""
    from oracle import pgm

    g = golden("psf_unmatched")
    rng = np.random.default_rng(8)
    C, H, W = g["images"].shape
    specs, ocomps = [], []
    for (h, w), (oy, ox) in (((21, 21), (-6, 13)), ((31, 31), (20, 40)), ((15, 23), (5, -4)),
                             ((11, 11), (35, 55))):
        morph = rng.random((h, w)).astype(np.float32)
        morph /= morph.max()
        sed = rng.uniform(0.5, 2, C).astype(np.float32)
        specs.append(amd.ComponentSpec(sed, morph, (oy, ox), sed_min_step=0.01))
        ocomps.append(pgm.Component(sed.copy(), morph.copy(), (oy, ox), sed_min_step=0.01))
    w = np.full(g["images"].shape, 0.25, dtype=np.float32)
    w[:, :3, :] = 0  # masked rows: log_norm must skip them
    batch = amd.BlendBatch(g["images"][None], w[None], [specs], kernel=g["diff_kernel"], max_iter=4,
                           conv_path=path)
    assert batch.fft_shape == ((75, 96) if path == "rocfft" else (64, 80))
    sc = pgm.Scene(g["images"].shape, g["images"], w, g["diff_kernel"], ocomps)
    model, rendered, logL = batch.forward()
    ref_model = sc.get_model()
    assert rel_err(model[0], ref_model) < RTOL
    assert rel_err(rendered[0], sc.render(ref_model)) < RTOL
    assert abs(logL[0] - sc.log_likelihood(sc.render(ref_model))) < RTOL * abs(logL[0])
    g_sed, g_morph = batch.gradient()
    _, grads = sc.loss_and_gradients()
    for k, ((gs, gm), (s_sed, s_morph)) in enumerate(zip(grads, grad_scales(sc))):
        assert np.abs(g_sed[k] - gs).max() < RTOL * s_sed, k
        assert np.abs(g_morph[k] - gm).max() < RTOL * s_morph, k
    # reference render of a fixed cube with the per-band kernel (golden from the reference)
    spec = [amd.ComponentSpec(np.eye(C, dtype=np.float32)[c], g["model"][c], (0, 0),
                              prox_flags=0) for c in range(C)]
    b2 = amd.BlendBatch(g["images"][None], w[None], [spec], kernel=g["diff_kernel"], max_iter=2,
                        conv_path=path)
    _, rendered, _ = b2.forward()
    assert np.abs(rendered[0] - g["rendered"]).max() < RTOL * np.abs(g["rendered"]).max()


@pytest.mark.parametrize("path", PATHS)
def test_convolution_properties_full_size(amd, path):
    """size-independent properties at the benchmark shape (5x128x128, F=180x180):
    linearity, and <A x, y> = <x, A^T y> with the gradient path as A^T"""
    from scarlet_amd import synthetic

    obs, model_psf, diff = synthetic.psfs()
    rng = np.random.default_rng(5)
    C, H, W, T = 5, 128, 128, 64
    x = rng.standard_normal((C, H, W)).astype(np.float32)
    y = rng.standard_normal((C, H, W)).astype(np.float32)
    eye = np.eye(C, dtype=np.float32)
    tiles = [(c, ty, tx) for c in range(C) for ty in range(0, H, T) for tx in range(0, W, T)]

    def specs(cube):
        # the cube as unit-sed components, one 64x64 tile each
        return [amd.ComponentSpec(eye[c], cube[c, ty:ty + T, tx:tx + T], (ty, tx), prox_flags=0)
                for c, ty, tx in tiles]

    def untile(parts):
        out = np.zeros((C, H, W), dtype=np.float64)
        for (c, ty, tx), p in zip(tiles, parts):
            out[c, ty:ty + T, tx:tx + T] = p
        return out

    def render(cube, data=None):
        d = np.zeros((1, C, H, W), np.float32) if data is None else data[None]
        b = amd.BlendBatch(d, np.ones((1, C, H, W), np.float32), [specs(cube)], kernel=diff,
                           max_iter=2, conv_path=path)
        return b, b.forward()[1][0]

    _, ax = render(x)
    _, ay = render(y)
    _, axy = render(x + 2 * y)
    assert rel_err(axy, ax.astype(np.float64) + 2.0 * ay) < RTOL
    # with unit weights the gradient wrt the tiles is A^T (A x - y)
    b, _ = render(x, data=y)
    _, g_morph = b.gradient()
    at_r = untile(g_morph)
    r = (ax - y).astype(np.float32)
    lhs = np.sum(r.astype(np.float64) * render(r)[1])   # <r, A r>
    rhs = np.sum(at_r * r)                               # <A^T r, r>
    assert abs(lhs - rhs) < 1e-4 * abs(lhs)


def test_convergence_freezes_blend_and_error_flag(amd, hsc):
    batch = hsc_batch(amd, hsc, max_iter=60)
    n_iter, _ = batch.fit(max_iter=60, e_rel=1e-2)  # loose: converges early
    assert n_iter[0] < 60
    sed0, _ = batch.parameters()
    batch.step(int(n_iter[0]), 3, e_rel=1e-2, check_convergence=True)  # frozen
    sed1, _ = batch.parameters()
    assert_array_equal(sed0, sed1)
    assert batch.status()[0] == 0
    # non-finite input -> ArithmeticError (model.py:153-165)
    bad = hsc_batch(amd, hsc, max_iter=12)
    seds, morphs = bad.parameters()
    seds[0, 0] = np.nan
    bad.set_parameters(seds, morphs)
    with pytest.raises(ArithmeticError):
        bad.fit(max_iter=10, e_rel=1e-4)


@pytest.mark.parametrize("n_blends", [3, 48, 128])
def test_loss_history_of_a_blend_that_goes_non_finite(amd, n_blends):
    """The loss of the iteration in which a blend's parameters become non-finite is recorded
    (Blend._callback appends it before the step, blend.py:294-299), whatever launch the
    bookkeeping rides in: workgroups of the update launch (<= 1024 components) or a launch of
    its own (1280 components), and whichever of them the hardware ran first."""
    from scarlet_amd import synthetic

    scenes = synthetic.make_batch(range(1234, 1234 + n_blends))
    kern = synthetic.psfs()
    comps = [[amd.ComponentSpec(s["seds"][k], s["morphs"][k], s["origins"][k],
                                sed_min_step=s["noise_rms"]) for k in range(len(s["morphs"]))]
             for s in scenes]
    victims = sorted({0, n_blends // 2, n_blends - 1})
    for trial in range(3):
        b = amd.BlendBatch(np.stack([s["data"] for s in scenes]),
                           np.stack([s["weights"] for s in scenes]), comps, kernel=kern[2],
                           max_iter=12)
        b.step(0, 3, e_rel=1e-3)
        seds, morphs = b.parameters()
        for j in victims:   # a different component of each victim
            seds[10 * j + (j + trial) % 10, 1] = np.nan
        b.set_parameters(seds, morphs)
        b.step(3, 4, e_rel=1e-3)
        n_active, first_bad = b.status()
        assert first_bad == victims[0] and n_active == n_blends - len(victims)
        states = b.states()
        lengths = [len(h) for h in b.loss_history()]
        for j in range(n_blends):
            assert states[j] == (3 if j in victims else 0), (trial, j)
            # iterations 0 .. 2, plus iteration 3 whose update failed; the others all seven
            assert lengths[j] == (4 if j in victims else 7), (trial, j, lengths[j])
        b.close()


def test_resize_on_the_device_equals_the_host_resize_and_blends_keep_their_own_counters(amd):
    """smi_batch_update_components with keep = 2 / 3 (the resize of ImageMorphology.update,
    morphology.py:132-207, carried out by the device: centred slice; zero-padded moments and
    np.pad(mode="linear_ramp") of the image in float64 -- numpy's two ramp formulas, rows
    rounded to float32 between the axes or not) against the same arrays made by numpy on the
    host; then smi_batch_set_iteration_base: blends 1 and 3 start their adaprox call anew at
    counter 11 while the others go on, in ONE launch per iteration -- the same bits as the
    two groups stepped in turn with the others paused."""
    from scarlet_amd import synthetic

    n_blends = 4
    scenes = synthetic.make_batch(range(4321, 4321 + n_blends))
    kern = synthetic.psfs()
    data = np.stack([s["data"] for s in scenes])
    weights = np.stack([s["weights"] for s in scenes])
    n = 10 * n_blends

    def specs_of(seds, morphs, origins, steps):
        return [[amd.ComponentSpec(seds[10 * i + k], morphs[10 * i + k], origins[10 * i + k],
                                   sed_min_step=scenes[i]["noise_rms"], morph_step=steps[10 * i + k])
                 for k in range(10)] for i in range(n_blends)]

    origins = [tuple(int(v) for v in s["origins"][k]) for s in scenes for k in range(10)]
    seds0 = [s["seds"][k] for s in scenes for k in range(10)]
    morphs0 = [s["morphs"][k].copy() for s in scenes for k in range(10)]
    # edge lines without a zero (the i * (edge / pad) branch of np.linspace) next to the usual
    # ones with zeros (the (i / pad) * edge branch)
    rng = np.random.default_rng(5)
    for k in (11, 14, 31):
        morphs0[k] = np.maximum(morphs0[k], rng.uniform(1e-4, 1e-2, morphs0[k].shape).astype(np.float32))
    steps = [1e-2] * n

    def start():
        b = amd.BlendBatch(data, weights, specs_of(seds0, morphs0, origins, steps), kernel=kern[2],
                           max_iter=40)
        b.step(0, 11, e_rel=1e-3)
        return b

    # what the device is asked to do: -5 shrink to 31^2, +5 / +10 grow to 51^2 / 61^2
    grow = {11: 5, 12: -5, 14: 10, 17: 5, 31: 5, 33: -5, 38: 5}
    wide = {14, 17, 38}  # host image float64: no rounding between the axes

    def image(a, k):
        d = grow.get(k, 0)
        if d < 0:
            return a[-d:d, -d:d]
        if d > 0:
            src = a.astype(np.float64) if k in wide else a
            return np.pad(src, d, mode="linear_ramp").astype(np.float32)
        return a

    def moment(a, k):
        d = grow.get(k, 0)
        return a[-d:d, -d:d] if d < 0 else np.pad(a, d) if d > 0 else a

    keep = np.ones(n, dtype=np.int32)
    rows = np.array(sorted(grow))
    keep[rows] = [3 if k in wide else 2 for k in rows]
    new_origin = np.array([[origins[k][0] - grow[k], origins[k][1] - grow[k]] for k in rows])
    new_size = np.array([41 + 2 * grow[k] for k in rows])

    def resize(batch):
        seds, morphs = batch.parameters()
        mom = batch.moments()
        batch.update_components(
            specs_of(seds0, morphs0, origins, steps), keep, [],
            resized=dict(rows=rows, origin_y=new_origin[:, 0], origin_x=new_origin[:, 1],
                         size=new_size, morph_step=np.full(rows.size, 5e-3)))
        s1, m1 = batch.parameters()
        assert_array_equal(s1, seds)
        got = batch.moments()
        for k in range(n):
            assert_array_equal(m1[k], image(morphs[k], k), err_msg="image %d" % k)
            for name in ("m_morph", "v_morph", "vhat_morph"):
                assert_array_equal(got[name][k], moment(mom[name][k], k), err_msg="%s %d" % (name, k))
        for name in ("m_sed", "v_sed", "vhat_sed"):
            assert_array_equal(got[name], mom[name])
        return seds, morphs, mom

    # images as they were set (edge lines without a zero: the other formula of the ramp) ...
    fresh = amd.BlendBatch(data, weights, specs_of(seds0, morphs0, origins, steps), kernel=kern[2],
                           max_iter=40)
    _, morphs, _ = resize(fresh)
    assert all((morphs[k][[0, -1]] != 0).all() and (morphs[k][:, [0, -1]] != 0).all()
               for k in (11, 14, 31))
    fresh.close()
    # ... and after eleven iterations, moments and all (edges with zeros)
    live = start()
    seds, morphs, mom = resize(live)
    assert (morphs[17][0] == 0).any()

    # blends 1 and 3 (the resized ones) restart at counter 11, all four in one launch ...
    base = np.array([0, 11, 0, 11], dtype=np.int32)
    live.set_iteration_base(base)
    live.step(11, 5, e_rel=1e-3, check_convergence=True)
    # ... against: the same state, the two groups stepped in turn with the others paused
    new_specs = specs_of(list(seds), [image(morphs[k], k) for k in range(n)],
                         [tuple(new_origin[list(rows).index(k)]) if k in grow else origins[k]
                          for k in range(n)],
                         [5e-3 if k in grow else 1e-2 for k in range(n)])
    turn = amd.BlendBatch(data, weights, new_specs, kernel=kern[2], max_iter=40)
    turn.set_moments(m_sed=mom["m_sed"], v_sed=mom["v_sed"], vhat_sed=mom["vhat_sed"],
                     m_morph=[moment(a, k) for k, a in enumerate(mom["m_morph"])],
                     v_morph=[moment(a, k) for k, a in enumerate(mom["v_morph"])],
                     vhat_morph=[moment(a, k) for k, a in enumerate(mom["vhat_morph"])])
    turn.set_previous_loss(np.array([h[10] for h in live.loss_history()]))
    turn.set_states(np.array([0, 2, 0, 2], dtype=np.int32))
    turn.step(11, 5, e_rel=1e-3, check_convergence=True)
    turn.set_states(np.array([2, 0, 2, 0], dtype=np.int32))
    turn.step(0, 5, e_rel=1e-3, check_convergence=True)
    s1, m1 = live.parameters()
    s2, m2 = turn.parameters()
    assert_array_equal(s1, s2)
    for a, b_ in zip(m1, m2):
        assert_array_equal(a, b_)
    for name, arrays in live.moments().items():
        for a, b_ in zip(arrays, turn.moments()[name]):
            assert_array_equal(a, b_)
    for a, b_ in zip(live.loss_history(), turn.loss_history()):
        assert_array_equal(a[11:16], b_[:5])
    live.close()
    turn.close()


def test_resize_on_a_live_batch_equals_a_rebuilt_batch(amd):
    """smi_batch_resize_test / smi_batch_get_component_states / smi_batch_update_components:
    the reductions of ImageMorphology.update (morphology.py:132-207) on the device against
    the host's expressions, state records against the full download, and a batch whose
    component table changed under it (one box shrunk, one grown, the others kept on the
    device; one blend paused) against a batch built anew from the same state: the same bits
    after further iterations."""
    from scarlet_amd import synthetic
    from scarlet_amd.morphology import _edge_pull, _empty_margin

    n_blends = 5
    scenes = synthetic.make_batch(range(1234, 1234 + n_blends))
    kern = synthetic.psfs()
    data = np.stack([s["data"] for s in scenes])
    weights = np.stack([s["weights"] for s in scenes])

    def specs_of(seds, morphs, origins, steps):
        return [[amd.ComponentSpec(seds[10 * i + k], morphs[10 * i + k], origins[10 * i + k],
                                   sed_min_step=scenes[i]["noise_rms"], morph_step=steps[10 * i + k])
                 for k in range(10)] for i in range(n_blends)]

    origins = [tuple(int(v) for v in s["origins"][k]) for s in scenes for k in range(10)]
    steps = [1e-2] * (10 * n_blends)
    seds0 = [s["seds"][k] for s in scenes for k in range(10)]
    morphs0 = [s["morphs"][k] for s in scenes for k in range(10)]
    live = amd.BlendBatch(data, weights, specs_of(seds0, morphs0, origins, steps), kernel=kern[2],
                          max_iter=40)
    live.step(0, 11, e_rel=1e-3)
    seds, morphs = live.parameters()
    mom = live.moments()

    # the device's reductions against the host's
    margin, pull = live.resize_test()
    for k in range(10 * n_blends):
        assert margin[k] == _empty_margin(morphs[k], 0), k
        want = np.nanmax(_edge_pull(morphs[k], mom["m_morph"][k].astype(np.float64),
                                    mom["v_morph"][k].astype(np.float64), 1e-2))
        assert abs(pull[k] - want) <= 1e-6 * abs(want) + 1e-300, (k, pull[k], want)
    # state records against the full download
    pick = [3, 17, 18, 49]
    for k, rec in zip(pick, live.component_states(pick)):
        assert_array_equal(rec["sed"], seds[k])
        assert_array_equal(rec["morph"], morphs[k])
        for name in ("m_sed", "v_sed", "vhat_sed"):
            assert_array_equal(rec[name], mom[name][k])
        for name in ("m_morph", "v_morph", "vhat_morph"):
            assert_array_equal(rec[name], mom[name][k])

    # component 12 shrinks to 31^2, component 37 grows to 51^2 (zero padding), steps halved
    def resized(a, k):
        if k == 12:
            return a[5:-5, 5:-5]
        if k == 37:
            return np.pad(a, 5)
        return a

    new_morphs = [resized(morphs[k], k) for k in range(10 * n_blends)]
    new_origins = list(origins)
    new_origins[12] = (origins[12][0] + 5, origins[12][1] + 5)
    new_origins[37] = (origins[37][0] - 5, origins[37][1] - 5)
    new_steps = list(steps)
    new_steps[12] = new_steps[37] = 5e-3
    new_specs = specs_of(list(seds), new_morphs, new_origins, new_steps)
    keep = np.ones(10 * n_blends, dtype=bool)
    keep[10:20] = False   # blend 1 goes over the host entirely, blend 3 only its resized box
    keep[37] = False
    records = [dict(sed=seds[k], m_sed=mom["m_sed"][k], v_sed=mom["v_sed"][k],
                    vhat_sed=mom["vhat_sed"][k], morph=new_morphs[k],
                    m_morph=resized(mom["m_morph"][k], k), v_morph=resized(mom["v_morph"][k], k),
                    vhat_morph=resized(mom["vhat_morph"][k], k))
               for k in np.flatnonzero(~keep)]
    live.update_components(new_specs, keep, records)
    paused = np.zeros(n_blends, dtype=np.int32)
    paused[2] = 2
    live.set_states(paused)
    live.step(11, 6, e_rel=1e-3)
    state, count = live.progress()
    assert list(count) == [17, 17, 11, 17, 17] and list(state) == [0, 0, 2, 0, 0]

    fresh = amd.BlendBatch(data, weights, new_specs, kernel=kern[2], max_iter=40)
    fresh.set_moments(m_sed=mom["m_sed"], v_sed=mom["v_sed"], vhat_sed=mom["vhat_sed"],
                      m_morph=[resized(a, k) for k, a in enumerate(mom["m_morph"])],
                      v_morph=[resized(a, k) for k, a in enumerate(mom["v_morph"])],
                      vhat_morph=[resized(a, k) for k, a in enumerate(mom["vhat_morph"])])
    fresh.set_states(paused)
    fresh.step(11, 6, e_rel=1e-3)
    s1, m1 = live.parameters()
    s2, m2 = fresh.parameters()
    assert_array_equal(s1, s2)
    for a, b_ in zip(m1, m2):
        assert_array_equal(a, b_)
    for name, got in live.moments().items():
        for a, b_ in zip(got, fresh.moments()[name]):
            assert_array_equal(a, b_)
    # the paused blend did not move, the others did
    assert_array_equal(s1[20:30], seds[20:30])
    assert not np.array_equal(s1[:10], seds[:10])
    new_loss = [h[11:] for h in live.loss_history()]
    for i, h in enumerate(fresh.loss_history()):
        assert_array_equal(new_loss[i], h)
    live.close()
    fresh.close()


def test_make_batch_matches_host_generator(amd):
    from scarlet_amd import synthetic

    kern = synthetic.psfs()
    gpu = synthetic.make_batch([1234, 77], kernel=kern)
    for s, seed in zip(gpu, (1234, 77)):
        host = synthetic.make_blend(seed, kernel=kern)
        assert np.abs(s["data"] - host["data"]).max() < 1e-5 * np.abs(host["data"]).max()
        assert_array_equal(s["seds"], host["seds"])
        assert_array_equal(np.array(s["morphs"]), np.array(host["morphs"]))
        assert_array_equal(s["origins"], host["origins"])


# ------------------------------------------------------------------ edge cases
def _random_scene(rng, C, H, W, boxes, kernel_shape=None, **comp_kw):
    from oracle import pgm

    specs, ocomps = [], []
    import scarlet_amd as amd

    for (h, w), (oy, ox) in boxes:
        yy, xx = np.mgrid[:h, :w]
        morph = np.exp(-0.5 * (((yy - h // 2) / (0.2 * h)) ** 2 + ((xx - w // 2) / (0.25 * w)) ** 2))
        morph = (morph * rng.uniform(0.7, 1.3, morph.shape)).astype(np.float32)
        morph /= morph.max()
        sed = rng.uniform(0.5, 3, C).astype(np.float32)
        specs.append((sed, morph, (oy, ox)))
    kernel = None
    if kernel_shape is not None:
        from scarlet_amd import fft
        from scarlet_amd.psf import GaussianPSF

        obs = GaussianPSF(1.8, boxsize=kernel_shape).get_model().astype(np.float32)
        mod = GaussianPSF(0.8).get_model().astype(np.float32)
        kernel = fft.match_psf(fft.Fourier(obs), fft.Fourier(mod), padding=10).image.astype(np.float32)
    truth = pgm.Scene((C, H, W), None, None, kernel,
                      [pgm.Component(s.copy(), m.copy(), o) for s, m, o in specs])
    data = truth.render(truth.get_model()) + rng.normal(0, 0.05, (C, H, W)).astype(np.float32)
    data = data.astype(np.float32)
    weights = np.full((C, H, W), 400.0, dtype=np.float32)
    return specs, kernel, data, weights


def _compare_steps(amd, specs, kernel, data, weights, n_it, comp_kw, ocomp_kw, flags=None,
                   oracle_attrs=None, **batch_kw):
    from oracle import pgm

    comps = [amd.ComponentSpec(s * 0.8, m, o, sed_min_step=0.05, **({"prox_flags": flags} if flags is not None else {}), **comp_kw)
             for s, m, o in specs]
    batch = amd.BlendBatch(data[None], weights[None], [comps], kernel=kernel, max_iter=n_it + 1, **batch_kw)
    sc = pgm.Scene(data.shape, data, weights, kernel,
                   [pgm.Component((s * 0.8).astype(np.float32), m.copy(), o, sed_min_step=0.05, **ocomp_kw)
                    for s, m, o in specs])
    for c in sc.components:
        for name, value in (oracle_attrs or {}).items():
            setattr(c, name, value)
    batch.step(0, n_it, e_rel=1e-3)
    for it in range(n_it):
        sc.step(it, 1e-3)
    sed, morphs = batch.parameters()
    assert_loss_close(batch.loss_history()[0], sc.loss, sc.log_norm, rtol=3e-5)
    for k, c in enumerate(sc.components):
        assert rel_err(sed[k], c.sed) < 2e-4, k
        assert np.abs(morphs[k] - c.morph).max() < 2e-4, k
    return batch, sc


@pytest.mark.parametrize("mode,g,symmetric", [("flat", 0.1, False), ("nearest", 0.0, False),
                                              ("angle", 0.25, True)])
def test_weightings_symmetry_and_gradient(amd, mode, g, symmetric):
    from scarlet_amd import _lib

    rng = np.random.default_rng(21)
    boxes = [((21, 21), (3, 5)), ((31, 31), (20, 25)), ((25, 35), (30, -4)), ((22, 30), (-5, 40))]
    specs, kernel, data, weights = _random_scene(rng, 3, 64, 72, boxes, kernel_shape=21)
    flags = _lib.PROX_EXTENDED_SOURCE | (_lib.PROX_SYMMETRY if symmetric else 0)
    _compare_steps(amd, specs, kernel, data, weights, 3,
                   dict(neighbor_weight=mode, min_gradient=g),
                   dict(monotonic=mode, min_gradient=g, symmetric=symmetric), flags=flags)


@pytest.mark.parametrize("kind,thresh,type", [("l1", 0.02, "absolute"), ("l0", 0.05, "absolute"),
                                              ("l1", 3.0, "relative"), ("l0", 8.0, "relative")])
def test_sparsity_constraints_and_center_floor(amd, kind, thresh, type):
    """L0Constraint / L1Constraint in the chain (absolute and relative thresholds; the
    relative one scales with the step of every proximal sub-iteration) and a
    CenterOnConstraint floor other than 1e-6"""
    from scarlet_amd import _lib

    rng = np.random.default_rng(31)
    boxes = [((21, 21), (3, 5)), ((31, 31), (20, 25)), ((25, 35), (30, -4))]
    specs, kernel, data, weights = _random_scene(rng, 3, 64, 72, boxes, kernel_shape=15)
    flags = _lib.PROX_EXTENDED_SOURCE | (_lib.PROX_L1 if kind == "l1" else _lib.PROX_L0)
    flags |= _lib.PROX_L_RELATIVE if type == "relative" else 0
    _compare_steps(amd, specs, kernel, data, weights, 3,
                   dict(l_thresh=thresh, center_floor=1e-3),
                   dict(sparsity=(kind, thresh, type), tiny=1e-3), flags=flags)


def test_many_bands_big_boxes_null_renderer(amd):
    """C = 10 (> one band chunk), boxes 61^2 (register path NPL=59) and 81^2 (generic
    LDS kernel), NullRenderer"""
    rng = np.random.default_rng(22)
    specs, kernel, data, weights = _random_scene(
        rng, 10, 96, 100, [((61, 61), (10, 20)), ((41, 41), (50, 50))])
    _compare_steps(amd, specs, None, data, weights, 2, {}, {})
    specs, kernel, data, weights = _random_scene(
        rng, 2, 120, 120, [((81, 81), (10, 20)), ((21, 21), (90, 90))])
    _compare_steps(amd, specs, None, data, weights, 2, {}, {})


@pytest.mark.parametrize("H,W,F", [
    (50, 85, (64, 96)), (85, 50, (96, 64)), (70, 110, (80, 128)), (110, 70, (128, 80)),
    (88, 150, (96, 160)), (150, 88, (160, 96)), (120, 120, (128, 128)), (57, 153, (64, 160)),
    (153, 57, (160, 64)), (89, 73, (96, 80)), (60, 121, (80, 128)), (40, 40, (64, 64)),
    (149, 141, (160, 160)), (103, 139, (128, 160)), (141, 101, (160, 128))])
def test_fused_path_every_fft_length(amd, H, W, F):
    """every LDS-resident FFT length (64, 80, 96, 128, 160 = 16 x {4,5,6,8,10}) on both
    axes: forward, gradient and two full steps against the oracle, 15^2 kernel.  The last three
    run in 1024-thread workgroups (the others in 512-thread ones); 149 rows are 75 row pairs,
    i.e. more stride-pass work items than a workgroup has threads, odd heights leave half a
    pair, widths off the multiples of 16 a block of columns that straddles the frame"""
    rng = np.random.default_rng(H * 1000 + W)
    boxes = [((21, 21), (3, 5)), ((31, 31), (H - 35, W - 36)), ((15, 25), (H // 2, -6)),
             ((25, 15), (-7, W // 2))]
    specs, kernel, data, weights = _random_scene(rng, 3, H, W, boxes, kernel_shape=15)
    batch, sc = _compare_steps(amd, specs, kernel, data, weights, 2, {}, {}, conv_path="fused")
    assert tuple(batch.fft_shape) == F


def test_fused_and_rocfft_paths_agree_on_random_scenes(amd):
    """12 random scenes (frame 30..150 px per side, 1..6 bands, odd kernels 5..43 px shared
    or per band, ragged boxes that may overhang the frame): the LDS-resident convolution
    and the rocFFT pipeline must give the same forward, gradients and one full step"""
    from scarlet_amd.psf import GaussianPSF
    from scarlet_amd import fft

    rng = np.random.default_rng(77)
    for case in range(12):
        C = int(rng.integers(1, 7))
        H, W = (int(v) for v in rng.integers(30, 151, 2))
        P = int(rng.integers(2, 22)) * 2 + 1
        if H + P // 2 > 160 or W + P // 2 > 160:
            P = 2 * min(160 - H, 160 - W) - 1
        per_band = bool(rng.integers(0, 2)) and C > 1
        sig = rng.uniform(1.2, 2.5, C if per_band else 1)
        obs = np.concatenate([GaussianPSF(float(sg), boxsize=P).get_model() for sg in sig]).astype(np.float32)
        mod = GaussianPSF(0.8).get_model().astype(np.float32)
        kernel = fft.match_psf(fft.Fourier(obs), fft.Fourier(mod), padding=10).image.astype(np.float32)
        specs = []
        for _ in range(int(rng.integers(1, 6))):
            h, w = (int(v) for v in rng.integers(5, 36, 2))
            oy, ox = int(rng.integers(-h // 2, H - h // 2)), int(rng.integers(-w // 2, W - w // 2))
            morph = rng.random((h, w)).astype(np.float32)
            specs.append((rng.uniform(0.5, 2, C).astype(np.float32), morph / morph.max(), (oy, ox)))
        data = rng.normal(0, 1, (C, H, W)).astype(np.float32)
        weights = rng.uniform(0.5, 2, (C, H, W)).astype(np.float32)
        out = []
        for path in PATHS:
            comps = [amd.ComponentSpec(s_, m_, o_, sed_min_step=0.01, prox_flags=_lib_flags())
                     for s_, m_, o_ in specs]
            b = amd.BlendBatch(data[None], weights[None], [comps], kernel=kernel, max_iter=3,
                               conv_path=path)
            model, rendered, logL = b.forward()
            g_sed, g_morph = b.gradient()
            b.step(0, 1, e_rel=1e-3)
            sed, morphs = b.parameters()
            out.append((rendered[0], logL[0], g_sed, np.concatenate([g.ravel() for g in g_morph]),
                        sed, np.concatenate([m.ravel() for m in morphs])))
            b.close()
        a, f = out
        scale = np.abs(a[0]).max()
        assert np.abs(a[0] - f[0]).max() < 2e-5 * scale, case
        assert abs(a[1] - f[1]) < 2e-5 * abs(a[1]), case
        for i in (2, 3):
            assert np.abs(a[i] - f[i]).max() < 1e-4 * max(np.abs(a[i]).max(), 1e-3), (case, i)
        assert np.abs(a[4] - f[4]).max() < 2e-4 * np.abs(a[4]).max(), case
        assert np.abs(a[5] - f[5]).max() < 2e-4, case


def _lib_flags():
    from scarlet_amd import _lib

    return _lib.PROX_POSITIVE | _lib.PROX_NORM_MAX


def test_batch_invariance_at_benchmark_shape(amd):
    """a blend of the benchmark workload (5 x 128 x 128, 10 components of 41 x 41) gives
    bit-identical losses and parameters whether it runs alone or inside a batch of 48"""
    from scarlet_amd import synthetic

    scenes = synthetic.make_batch(range(1234, 1234 + 48))
    kern = synthetic.psfs()

    def run(sel):
        comps = [[amd.ComponentSpec(s["seds"][k], s["morphs"][k], s["origins"][k],
                                    sed_min_step=s["noise_rms"]) for k in range(len(s["morphs"]))]
                 for s in sel]
        b = amd.BlendBatch(np.stack([s["data"] for s in sel]), np.stack([s["weights"] for s in sel]),
                           comps, kernel=kern[2], max_iter=8)
        b.step(0, 6, e_rel=1e-3)
        out = b.loss_history(), b.parameters()
        b.close()
        return out

    loss_all, (sed_all, morph_all) = run(scenes)
    for i in (0, 17, 47):
        loss_one, (sed_one, morph_one) = run([scenes[i]])
        assert_array_equal(loss_one[0], loss_all[i])
        assert_array_equal(sed_one, sed_all[10 * i:10 * i + 10])
        for a, b_ in zip(morph_one, morph_all[10 * i:10 * i + 10]):
            assert_array_equal(a, b_)


def test_config3_at_its_own_size_follows_the_oracle(amd):
    """BASELINE configs[2] at the size bench.py runs it: one batch of 1024 blends, and rank 7's
    shard of an 8-GPU job (blends 896 .. 1023) as a batch of its own.  Twenty iterations from
    the initial parameters; the loss histories of blends 0, 511 and 1023 against the oracle
    (2e-5 relative on chi^2 over the first twelve iterations, 5e-4 through the transient
    later on: the tolerances of the whole-fit tests), and the shard's blends bit for bit the
    same as inside the full batch."""
    from oracle import pgm
    from scarlet_amd import synthetic

    n_total, n_it = 1024, 20
    kern = synthetic.psfs()
    scenes = synthetic.make_batch(range(1234, 1234 + n_total), kernel=kern)

    def fit(sel):
        comps = [[amd.ComponentSpec(s["seds"][k], s["morphs"][k], s["origins"][k],
                                    sed_min_step=s["noise_rms"]) for k in range(len(s["morphs"]))]
                 for s in sel]
        b = amd.BlendBatch(np.stack([s["data"] for s in sel]), np.stack([s["weights"] for s in sel]),
                           comps, kernel=kern[2], max_iter=n_it + 1)
        b.step(0, n_it, e_rel=1e-3, check_convergence=False)
        active, err = b.status()
        assert err < 0
        out = [np.array(l) for l in b.loss_history()], b.parameters()
        b.close()
        return out

    loss, (sed, morphs) = fit(scenes)
    assert len(loss) == n_total and all(len(l) == n_it for l in loss)
    for i in (0, 511, 1023):
        s = scenes[i]
        sc = pgm.Scene(s["data"].shape, s["data"], s["weights"], s["diff_kernel"],
                       [pgm.Component(s["seds"][k].copy(), s["morphs"][k].copy(), s["origins"][k],
                                      sed_min_step=s["noise_rms"]) for k in range(10)])
        for it in range(n_it):
            sc.step(it, 1e-3)
        assert_loss_close(loss[i][:12], sc.loss[:12], sc.log_norm, rtol=2e-5)
        assert_loss_close(loss[i], sc.loss, sc.log_norm, rtol=5e-4)
        for k, c in enumerate(sc.components):
            assert rel_err(sed[10 * i + k], c.sed) < 1e-3
            assert np.abs(morphs[10 * i + k] - c.morph).max() < 1e-3
    lo = 896  # dist.shard_range(1024, 7, 8)
    from scarlet_amd import dist

    assert dist.shard_range(n_total, 7, 8) == (lo, n_total)
    loss_s, (sed_s, morphs_s) = fit(scenes[lo:])
    for j in (0, 63, 127):
        assert_array_equal(loss_s[j], loss[lo + j])
        assert_array_equal(sed_s[10 * j:10 * j + 10], sed[10 * (lo + j):10 * (lo + j) + 10])
        for a, b_ in zip(morphs_s[10 * j:10 * j + 10], morphs[10 * (lo + j):10 * (lo + j) + 10]):
            assert_array_equal(a, b_)


def test_sub_ranges_on_streams_do_not_change_results(amd):
    """smi_batch_set_sub_ranges: ranges of blends stepped on streams of their own give
    bit-identical losses, iteration counts and parameters for every number of ranges,
    ragged blends (different component counts, an empty one) and convergence freezing
    included; the automatic choice is 3 ranges from 128 blends on (4 below 768 blends with eight HIP
    hardware queues)."""
    from scarlet_amd import synthetic

    kern = synthetic.psfs()
    scenes = synthetic.make_batch(range(900, 937), kernel=kern)
    keep = [10 - (i % 4) if i != 5 else 0 for i in range(len(scenes))]

    def run(n_sub):
        comps = [[amd.ComponentSpec(s["seds"][k], s["morphs"][k], s["origins"][k],
                                    sed_min_step=s["noise_rms"]) for k in range(keep[i])]
                 for i, s in enumerate(scenes)]
        b = amd.BlendBatch(np.stack([s["data"] for s in scenes]),
                           np.stack([s["weights"] for s in scenes]), comps, kernel=kern[2],
                           max_iter=60)
        b.set_sub_ranges(n_sub)
        assert b.sub_ranges() == max(n_sub, 1)
        n_iter, logL = b.fit(max_iter=60, e_rel=1e-3)
        out = n_iter, logL, b.loss_history(), b.parameters(), b.moments()
        b.close()
        return out

    ref = run(1)
    assert len(set(ref[0].tolist())) > 2  # blends stop at different iterations
    for n_sub in (2, 3, 7, 37):
        got = run(n_sub)
        assert_array_equal(got[0], ref[0])
        assert_array_equal(got[1], ref[1])
        for a, b_ in zip(got[2], ref[2]):
            assert_array_equal(a, b_)
        assert_array_equal(got[3][0], ref[3][0])
        for a, b_ in zip(got[3][1], ref[3][1]):
            assert_array_equal(a, b_)
        for key, want in ref[4].items():
            if isinstance(want, np.ndarray):
                assert_array_equal(got[4][key], want)
            else:
                for a, b_ in zip(got[4][key], want):
                    assert_array_equal(a, b_)
    # the automatic choice at the benchmark's scale: three ranges from 128 blends on
    many = synthetic.make_batch(range(3000, 3256), kernel=kern)
    hist = []
    for n_sub in (0, 1):
        comps = [[amd.ComponentSpec(s["seds"][k], s["morphs"][k], s["origins"][k],
                                    sed_min_step=s["noise_rms"]) for k in range(10)] for s in many]
        b = amd.BlendBatch(np.stack([s["data"] for s in many]),
                           np.stack([s["weights"] for s in many]), comps, kernel=kern[2],
                           max_iter=5)
        b.set_sub_ranges(n_sub)
        # (four ranges for fewer than 768 blends where HIP has eight hardware queues)
        auto = 4 if int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) >= 8 else 3
        assert b.sub_ranges() == (auto if n_sub == 0 else 1)
        b.step(0, 4, e_rel=1e-3)
        hist.append(np.array(b.loss_history()))
        b.close()
    assert_array_equal(hist[0], hist[1])
    small = amd.BlendBatch(scenes[0]["data"][None], scenes[0]["weights"][None],
                           [[amd.ComponentSpec(scenes[0]["seds"][0], scenes[0]["morphs"][0],
                                               scenes[0]["origins"][0])]], kernel=kern[2], max_iter=2)
    assert small.sub_ranges() == 1
    small.close()


def test_mixed_and_per_class_update_launches_agree(amd, hsc):
    """Components of several box-size classes: up to kMixedUpdateLimit (3072) components per
    range go through update_kernel_mixed (one launch, every wavefront runs the code of its
    class), larger ranges through one update_kernel_reg launch per class.  Both must give
    every blend the same bits: 320 copies of the quickstart blend (3200 components, boxes
    21^2 .. 61^2) in one range against a batch of three copies."""
    def run(nb):
        n = int(hsc["n_comp"])
        comps = [amd.ComponentSpec(hsc["sed_%d" % k], hsc["morph_%d" % k], hsc["origin_%d" % k],
                                   sed_min_step=hsc["min_step_%d" % k]) for k in range(n)]
        b = amd.BlendBatch(np.repeat(hsc["images"][None], nb, 0), np.repeat(hsc["weights"][None], nb, 0),
                           [comps] * nb, kernel=hsc["diff_kernel"], max_iter=16)
        b.set_sub_ranges(1)
        b.step(0, 12, e_rel=1e-3)
        out = b.loss_history(), b.parameters(), b.moments()
        b.close()
        return out

    big, small = run(320), run(3)
    n = int(hsc["n_comp"])
    for i in (0, 171, 319):
        assert_array_equal(big[0][i], small[0][0])
        assert_array_equal(big[1][0][i * n:(i + 1) * n], small[1][0][:n])
        for a, b_ in zip(big[1][1][i * n:(i + 1) * n], small[1][1][:n]):
            assert_array_equal(a, b_)
        for key in ("m_morph", "v_morph", "vhat_morph"):
            for a, b_ in zip(big[2][key][i * n:(i + 1) * n], small[2][key][:n]):
                assert_array_equal(a, b_)


def test_a_component_does_not_notice_its_batch_mates(amd, hsc):
    """A blend's bits do not depend on what else is in the batch: with a ConstraintChain(repeat)
    component somewhere in the batch every component goes through the general update kernel
    instead of the register-resident ones; the quickstart blend next to such a blend must
    come out exactly as it does alone."""
    n = int(hsc["n_comp"])

    def comps(repeat):
        return [amd.ComponentSpec(hsc["sed_%d" % k], hsc["morph_%d" % k], hsc["origin_%d" % k],
                                  sed_min_step=hsc["min_step_%d" % k],
                                  chain_repeat=2 if (repeat and k == 0) else 1) for k in range(n)]

    def run(specs):
        nb = len(specs)
        b = amd.BlendBatch(np.repeat(hsc["images"][None], nb, 0), np.repeat(hsc["weights"][None], nb, 0),
                           specs, kernel=hsc["diff_kernel"], max_iter=12)
        b.step(0, 10, e_rel=1e-3)
        out = b.loss_history(), b.parameters(), b.moments()
        b.close()
        return out

    alone, mixed = run([comps(False)]), run([comps(False), comps(True)])
    assert_array_equal(alone[0][0], mixed[0][0])
    assert_array_equal(alone[1][0], mixed[1][0][:n])
    for a, b_ in zip(alone[1][1], mixed[1][1][:n]):
        assert_array_equal(a, b_)
    for key in ("m_morph", "v_morph", "vhat_morph"):
        for a, b_ in zip(alone[2][key], mixed[2][key][:n]):
            assert_array_equal(a, b_)
    assert not np.array_equal(mixed[0][0], mixed[0][1])  # the repeated chain does something


def test_tiny_frames_and_single_band(amd):
    """frames much smaller than a chunk of the fused kernel, one band, boxes larger than
    the frame"""
    rng = np.random.default_rng(43)
    for (H, W), ks in (((8, 9), 5), ((17, 5), 3), ((3, 33), 7)):
        boxes = [((5, 5), (1, 2)), ((11, 13), (-4, -5))]
        specs, kernel, data, weights = _random_scene(rng, 1, H, W, boxes, kernel_shape=ks)
        for path in PATHS:
            _compare_steps(amd, specs, kernel, data, weights, 2, {}, {}, conv_path=path)


def test_more_than_64_components_in_one_blend(amd):
    """the render stage of the fused kernel walks the components of a blend in groups of
    64 (one component per lane of metadata): 70 small overlapping boxes"""
    rng = np.random.default_rng(41)
    boxes = [((9 + 2 * (k % 3), 9 + 2 * (k % 4)), (int(rng.integers(-3, 50)), int(rng.integers(-3, 60))))
             for k in range(70)]
    specs, kernel, data, weights = _random_scene(rng, 2, 56, 66, boxes, kernel_shape=11)
    for path in PATHS:
        _compare_steps(amd, specs, kernel, data, weights, 2, {}, {}, conv_path=path)


def test_boxes_beyond_the_lds(amd):
    """121^2 box (what lite's detection-image initialisation produces on small frames):
    the generic update kernel keeps x / psi / z in a global scratch area, the swept
    image in LDS"""
    rng = np.random.default_rng(24)
    specs, kernel, data, weights = _random_scene(
        rng, 2, 100, 110, [((121, 121), (-10, -6)), ((21, 21), (40, 70))])
    _compare_steps(amd, specs, None, data, weights, 2, {}, {})


def test_large_frame_uses_rocfft(amd):
    """200x180 frame + 25^2 kernel: the padded band does not fit the LDS, so the batch
    falls back to the rocFFT pipeline with the reference's FFT shape"""
    rng = np.random.default_rng(23)
    specs, kernel, data, weights = _random_scene(
        rng, 2, 200, 180, [((41, 41), (30, 40)), ((31, 31), (150, 120))], kernel_shape=25)
    batch, sc = _compare_steps(amd, specs, kernel, data, weights, 2, {}, {})
    from oracle import fftconv

    assert list(batch.fft_shape) == fftconv.fft_shape((2, 200, 180), kernel.shape, 3, (1, 2))
    with pytest.raises(Exception):
        amd.BlendBatch(data[None], weights[None], [[]], kernel=kernel, conv_path="fused")


def test_rocfft_shapes_with_transposed_partners(amd):
    """rocFFT 7.2 returns wrong transforms from a plan made while a plan of the transposed
    complex shape is alive ((Fy, Fx) next to (Fx / 2, 2 Fy); tools/rocfft_repro).  The
    library keeps plans of one FFT shape only between batches and moves an automatically
    chosen shape out of the way of a live partner: both orders, one after the other and
    alive at the same time, must agree with the oracle."""
    from oracle import pgm

    rng = np.random.default_rng(77)

    def scene(C, H, W, p):
        yy, xx = np.mgrid[:p, :p] - p // 2
        kernel = np.exp(-(yy**2 + xx**2) / 2.0)[None].astype(np.float32)
        kernel /= kernel.sum()
        data = rng.normal(0, 1, (C, H, W)).astype(np.float32)
        weights = np.ones((C, H, W), np.float32)
        morph = rng.random((9, 9)).astype(np.float32)
        sed = rng.uniform(0.5, 2, C).astype(np.float32)
        batch = amd.BlendBatch(data[None], weights[None], [[amd.ComponentSpec(sed, morph, (2, 12))]],
                               kernel=kernel, max_iter=2, conv_path="rocfft")
        sc = pgm.Scene((C, H, W), data, weights, kernel, [pgm.Component(sed.copy(), morph.copy(), (2, 12))])
        return batch, sc.render(sc.get_model())

    def check(batch, ref):
        _, rendered, _ = batch.forward()
        assert rel_err(rendered[0], ref) < RTOL, batch.fft_shape

    wide, square = (2, 14, 104, 13), (2, 54, 54, 3)  # FFT shapes (30, 120) and (60, 60)
    for first, second in ((wide, square), (square, wide)):
        a, ref_a = scene(*first)
        check(a, ref_a)
        a.close()
        b, ref_b = scene(*second)  # after the other shape: plans of one shape only are kept
        check(b, ref_b)
        b.close()
    for first, second in ((wide, square), (square, wide)):
        a, ref_a = scene(*first)
        b, ref_b = scene(*second)  # while the other is alive: steps aside
        assert a.fft_shape in ((30, 120), (60, 60)) and b.fft_shape not in ((30, 120), (60, 60))
        check(b, ref_b)
        check(a, ref_a)
        a.close()
        b.close()
    # an explicit shape cannot step aside: refused instead of silently wrong
    a, _ = scene(*wide)
    with pytest.raises(Exception, match="transposed"):
        amd.BlendBatch(np.zeros((1, 2, 54, 54), np.float32), np.ones((1, 2, 54, 54), np.float32),
                       [[amd.ComponentSpec(np.ones(2, np.float32), np.ones((5, 5), np.float32), (3, 3))]],
                       kernel=np.ones((1, 3, 3), np.float32) / 9, max_iter=2, conv_path="rocfft",
                       fft_shape=(60, 60))
    a.close()


def test_blend_without_components_and_mixed_batch(amd):
    """a batch where one blend has no components at all"""
    from oracle import pgm
    from scarlet_amd import synthetic

    kern = synthetic.psfs()
    s = synthetic.make_blend(5, kernel=kern)
    comps = [amd.ComponentSpec(s["seds"][k], s["morphs"][k], s["origins"][k],
                               sed_min_step=s["noise_rms"]) for k in range(10)]
    data = np.stack([s["data"], s["data"]])
    weights = np.stack([s["weights"], s["weights"]])
    batch = amd.BlendBatch(data, weights, [[], comps], kernel=kern[2], max_iter=4)
    model, rendered, logL = batch.forward()
    assert not model[0].any() and not rendered[0].any()
    sc = pgm.Scene(s["data"].shape, s["data"], s["weights"], kern[2], [])
    assert abs(logL[0] - sc.log_likelihood(np.zeros_like(s["data"]))) < 1e-5 * abs(logL[0])
    batch.step(0, 2)
    assert batch.status() == (2, -1)
    sc1 = pgm.Scene(s["data"].shape, s["data"], s["weights"], kern[2],
                    [pgm.Component(s["seds"][k].copy(), s["morphs"][k].copy(), s["origins"][k],
                                   sed_min_step=s["noise_rms"]) for k in range(10)])
    for it in range(2):
        sc1.step(it, 1e-3)
    assert_loss_close(batch.loss_history()[1], sc1.loss, sc1.log_norm)
    batch.close()
    # a batch without any component at all (found by tools/fuzz_batches.py)
    empty = amd.BlendBatch(data, weights, [[], []], kernel=kern[2], max_iter=4)
    _, rendered, logL2 = empty.forward()
    assert not rendered.any() and abs(logL2[1] - logL[0]) < 1e-5 * abs(logL[0])
    empty.step(0, 3)
    assert len(empty.loss_history()[0]) == 3 and empty.status()[1] == -1
    seds, morphs = empty.parameters()
    assert seds.shape == (0, 5) and morphs == []
    empty.close()


def test_batch_fit_to_convergence_mixed_states(amd):
    """64 blends fitted together until each one converges on its own: blends freeze at
    different iterations; two of them are compared with the oracle run alone"""
    from oracle import pgm
    from scarlet_amd import synthetic

    kern = synthetic.psfs()
    scenes = synthetic.make_batch(range(500, 564), kernel=kern)
    comps = [[amd.ComponentSpec(s["seds"][k], s["morphs"][k], s["origins"][k],
                                sed_min_step=s["noise_rms"]) for k in range(10)] for s in scenes]
    batch = amd.BlendBatch(np.stack([s["data"] for s in scenes]),
                           np.stack([s["weights"] for s in scenes]), comps, kernel=kern[2],
                           max_iter=200)
    n_iter, logL = batch.fit(max_iter=200, e_rel=1e-3)
    assert batch.status() == (0, -1)
    assert n_iter.min() >= 3 and n_iter.max() < 200 and len(set(n_iter.tolist())) > 3
    losses = batch.loss_history()
    for b in range(64):
        assert len(losses[b]) == n_iter[b]
        assert -losses[b][-1] > -losses[b][0]
        # the stopping rule held exactly at the last iteration and not before (blend.py:294-299)
        d = np.abs(np.diff(losses[b]))
        rel = d / np.abs(losses[b][1:])
        assert rel[-1] < 1e-3
        assert np.all(rel[1:-1] >= 1e-3)
    for b in (0, 37):
        s = scenes[b]
        sc = pgm.Scene(s["data"].shape, s["data"], s["weights"], kern[2],
                       [pgm.Component(s["seds"][k].copy(), s["morphs"][k].copy(), s["origins"][k],
                                      sed_min_step=s["noise_rms"]) for k in range(10)])
        n_ref, logL_ref = sc.fit(max_iter=200, e_rel=1e-3)
        assert abs(int(n_iter[b]) - n_ref) <= max(2, n_ref // 10)
        chi, chi_ref = -logL[b] - sc.log_norm, -logL_ref - sc.log_norm
        assert abs(chi - chi_ref) < 3e-3 * abs(chi_ref)


def test_saved_state_restarts_the_fit_bit_for_bit(amd):
    """smi_batch_save_state / smi_batch_restore_state (the warm restart of blend.py:155-170
    kept on the device): a fit repeated from the saved state gives the same losses,
    parameters and moments bit for bit, for image components and for point sources"""
    from conftest import golden
    from scarlet_amd import synthetic

    kern = synthetic.psfs()
    scenes = synthetic.make_batch(range(900, 904), kernel=kern)
    comps = [[amd.ComponentSpec(s["seds"][k], s["morphs"][k], s["origins"][k],
                                sed_min_step=s["noise_rms"]) for k in range(10)] for s in scenes]
    extended = amd.BlendBatch(np.stack([s["data"] for s in scenes]),
                              np.stack([s["weights"] for s in scenes]), comps, kernel=kern[2],
                              max_iter=12)
    for batch in (extended, _point_batch(amd, golden("point_source"), max_iter=12)):
        with pytest.raises(RuntimeError, match="no saved state"):
            batch.restore_state()
        batch.step(0, 3, e_rel=1e-3)  # a state with non-trivial moments
        batch.save_state()
        seds0, morphs0 = batch.parameters()
        runs = []
        for _ in range(2):
            batch.step(3, 6, e_rel=1e-3)
            runs.append((batch.loss_history(), batch.parameters(), batch.moments(),
                         batch.centers()))
            batch.restore_state()
            seds, morphs = batch.parameters()
            assert np.array_equal(seds, seds0)
            assert all(np.array_equal(a, b) for a, b in zip(morphs, morphs0))
            assert all(len(l) == 0 for l in batch.loss_history())
        (l0, p0, m0, c0), (l1, p1, m1, c1) = runs
        # the first run's history still holds the three iterations before the save
        assert all(len(a) == 9 and len(b) == 6 and np.array_equal(a[3:], b) for a, b in zip(l0, l1))
        assert np.array_equal(p0[0], p1[0])
        assert all(np.array_equal(a, b) for a, b in zip(p0[1], p1[1]))
        for name in m0:
            if name.endswith("sed"):
                assert np.array_equal(m0[name], m1[name])
            else:
                assert all(np.array_equal(x, y) for x, y in zip(m0[name], m1[name]))
        assert all(np.array_equal(c0[k], c1[k]) for k in c0)
        batch.close()


def test_kernel_shift_on_the_device(amd, hsc):
    """smi_batch_set_kernel_shift (ConvolutionRenderer(psf_shift=...), renderer.py:175-177,
    215-228): the shifted stamps and the gradient w.r.t. the shift against the oracle
    (whose gradient the golden pins to finite differences of the reference's forward), a
    zero shift reproduces the plain kernel, and steps follow the oracle; per-blend kernels
    give every blend a shift of its own"""
    from conftest import golden, hsc_scene
    from oracle import fftconv

    gp = golden("hsc_psf_shift")
    kernel = hsc["diff_kernel"]
    fft_shape = list(fftconv.fft_shape(kernel.shape, kernel.shape, padding=10, axes=(-2, -1)))
    shift0 = gp["psf_shift"].copy()

    def scene():
        sc = hsc_scene(hsc)
        for c in sc.components:
            c.source = None
        sc.psf_shift = shift0.copy()
        return sc

    sc = scene()
    batch = hsc_batch(amd, hsc, max_iter=10)
    batch.set_kernel_shift(kernel, shift0, fft_shape, step=1e-2)
    state = batch.kernel_shift(kernel=True)
    assert_array_equal(state["shift"][0], shift0)
    assert np.abs(state["kernel"][0] - sc.shifted_kernel()).max() < 2e-7 * np.abs(kernel).max()
    _, rendered, logL = batch.forward(model=False)
    model = sc.get_model()
    want = sc.render(model)
    assert np.abs(rendered[0] - want).max() < 1e-5 * np.abs(want).max()
    batch.gradient()
    g_ref = sc.psf_shift_gradient(model, want)
    assert_allclose(batch.kernel_shift()["gradient"][0], g_ref, rtol=2e-4)
    n_it = 8
    batch.step(0, n_it, e_rel=1e-3)
    for it in range(n_it):
        sc.step(it, 1e-3)
    state = batch.kernel_shift()
    assert np.abs(state["shift"][0] - sc.psf_shift).max() < 2e-5
    assert np.abs(state["shift"][0] - shift0).max() > 1e-3
    assert_allclose(state["m"][0], sc.m_psf, rtol=2e-3, atol=1e-3 * np.abs(sc.m_psf).max())
    assert_loss_close(batch.loss_history()[0], sc.loss, sc.log_norm, rtol=5e-5)
    # warm start from that state, restore of a saved state, and a fixed kernel again
    batch.save_state()
    batch.step(n_it, 2, e_rel=1e-3)
    after = batch.kernel_shift(kernel=True)
    batch.restore_state()
    back = batch.kernel_shift()
    assert_array_equal(back["shift"], state["shift"])
    batch.step(n_it, 2, e_rel=1e-3)
    again = batch.kernel_shift(kernel=True)
    assert_array_equal(again["shift"], after["shift"])
    assert_array_equal(again["kernel"], after["kernel"])
    batch.set_kernel(kernel)
    with pytest.raises(RuntimeError, match="no free kernel shift"):
        batch.kernel_shift()
    batch.close()

    # zero shift = the plain kernel (the Toeplitz maps are the identity)
    plain = hsc_batch(amd, hsc, max_iter=4)
    moved = hsc_batch(amd, hsc, max_iter=4)
    moved.set_kernel_shift(kernel, np.zeros(2), fft_shape, step=0.0)
    assert np.abs(moved.kernel_shift(kernel=True)["kernel"][0] - kernel).max() < 1e-7 * np.abs(kernel).max()
    plain.step(0, 3, e_rel=1e-3)
    moved.step(0, 3, e_rel=1e-3)
    assert_allclose(moved.loss_history()[0], plain.loss_history()[0], rtol=1e-6)
    plain.close()
    moved.close()

    # two blends with kernels of their own: each follows its own shift
    g = hsc
    specs = [amd.ComponentSpec(g["sed_%d" % k], g["morph_%d" % k], g["origin_%d" % k],
                               sed_min_step=g["min_step_%d" % k]) for k in range(int(g["n_comp"]))]
    two = amd.BlendBatch(np.stack([g["images"]] * 2), np.stack([g["weights"]] * 2), [specs, specs],
                         kernel=np.stack([kernel, kernel]), max_iter=10)
    shifts = np.stack([shift0, -0.5 * shift0])
    two.set_kernel_shift(np.stack([kernel, kernel]), shifts, fft_shape, step=1e-2)
    two.step(0, n_it, e_rel=1e-3)
    got = two.kernel_shift()["shift"]
    assert np.abs(got[0] - sc.psf_shift).max() < 2e-5
    sc2 = scene()
    sc2.psf_shift = shifts[1].copy()
    for it in range(n_it):
        sc2.step(it, 1e-3)
    assert np.abs(got[1] - sc2.psf_shift).max() < 2e-5
    assert_loss_close(two.loss_history()[1], sc2.loss, sc2.log_norm, rtol=5e-5)
    two.close()


# ---------------------------------------------------------------- point sources
POINT_SCENES = ["point_source", "point_source_moffat"]  # GaussianPSF / MoffatPSF model PSF


def _point_batch(amd, g, **kw):
    psf = dict(psf_sigma=0.9) if "moffat" not in g else dict(
        psf_sigma=float(g["moffat"][0]), psf_beta=float(g["moffat"][1]), boxsize=15)
    specs = []
    for k in range(int(g["n_src"])):
        if g["is_star"][k]:
            specs.append(amd.PointSourceSpec(g["sed_%d" % k], g["center_%d" % k],
                                             sed_min_step=g["min_step_%d" % k], **psf))
        else:
            specs.append(amd.ComponentSpec(g["sed_%d" % k], g["morph_%d" % k], g["origin_%d" % k],
                                           sed_min_step=g["min_step_%d" % k]))
    w = np.full(g["images"].shape, 0.25, dtype=np.float32)
    return amd.BlendBatch(g["images"][None], w[None], [specs], kernel=g["diff_kernel"], **kw)


@pytest.mark.parametrize("scene", POINT_SCENES)
@pytest.mark.parametrize("path", PATHS)
def test_point_source_scene_forward_and_gradient(amd, path, scene):
    """docs/tutorials/point_source.ipynb scene (3 PointSources + 2 ExtendedSources):
    PSF morphologies, model, rendered image and logL against the reference's golden
    values; gradients (centres included) against the oracle"""
    from conftest import point_scene

    g = golden(scene)
    batch = _point_batch(amd, g, max_iter=4, conv_path=path)
    sc = point_scene(g)
    _, morphs = batch.parameters()
    for k in range(int(g["n_src"])):
        if g["is_star"][k]:
            assert morphs[k].shape == g["morph_%d" % k].shape
            assert np.abs(morphs[k] - g["morph_%d" % k]).max() < 1e-7
    assert_allclose(batch.centers()["center"][g["is_star"]],
                    [g["center_%d" % k] for k in np.flatnonzero(g["is_star"])], rtol=0, atol=1e-12)
    model, rendered, logL = batch.forward()
    assert rel_err(model[0], g["model"]) < RTOL
    assert rel_err(rendered[0], g["rendered"]) < RTOL
    assert abs(logL[0] - float(g["logL"])) < RTOL * abs(float(g["logL"]))
    g_sed, g_morph = batch.gradient()
    g_center = batch.centers()["gradient"]
    _, grads = sc.loss_and_gradients()
    for k, c in enumerate(sc.components):
        s_sed = np.abs(grads[k][0]).max()
        assert np.abs(g_sed[k] - grads[k][0]).max() < 2e-5 * max(s_sed, 1.0), k
        if g["is_star"][k]:
            assert np.abs(g_center[k] - grads[k][1]).max() < 2e-5 * np.abs(grads[k][1]).max() + 1e-3, k
        else:
            assert np.abs(g_morph[k] - grads[k][1]).max() < 2e-5 * np.abs(grads[k][1]).max() + 1e-3, k


@pytest.mark.parametrize("scene", POINT_SCENES)
@pytest.mark.parametrize("path", PATHS)
def test_point_source_scene_steps(amd, path, scene):
    """12 full iterations of the mixed scene: losses, spectra, centres, morphologies"""
    from conftest import point_scene

    g = golden(scene)
    n_it = 12
    batch = _point_batch(amd, g, max_iter=n_it + 1, conv_path=path)
    sc = point_scene(g)
    batch.step(0, n_it, e_rel=1e-4)
    for it in range(n_it):
        sc.step(it, 1e-4)
    assert_loss_close(batch.loss_history()[0], sc.loss, sc.log_norm, rtol=1e-4)
    sed, morphs = batch.parameters()
    ctr = batch.centers()
    for k, c in enumerate(sc.components):
        assert rel_err(sed[k], c.sed) < 5e-4, k
        assert np.abs(morphs[k] - c.morph).max() < 5e-4, k
        if g["is_star"][k]:
            assert np.abs(ctr["center"][k] - c.center).max() < 1e-4, k
            assert_allclose(ctr["m"][k], c.m_center, rtol=1e-3, atol=1e-3 * np.abs(c.m_center).max())
            assert_allclose(ctr["vhat"][k], c.vhat_center, rtol=2e-3)
    # the centres did move
    assert max(np.abs(c.center - g["center_%d" % k]).max()
               for k, c in enumerate(sc.components) if g["is_star"][k]) > 1e-3


@pytest.mark.parametrize("path", PATHS)
def test_point_sources_on_an_image_psf(amd, path):
    """PointSource on an ImagePSF model PSF (psf.py:205-234: the stored image Fourier-shifted
    to the centre): on the device a component with a fixed image and a free Fourier shift =
    centre - mean(box bounds).  Shifted stamps, model, rendered image and logL against the
    reference's golden values; gradients (centres included) and twelve full iterations against
    the oracle; the whole fit stops where the oracle's does."""
    from conftest import point_scene

    g = golden("point_source_image")
    stars = np.flatnonzero(g["is_star"])
    stamp = g["psf_image"].astype(np.float32)
    box_center = {k: np.asarray(g["origin_%d" % k], dtype=np.float64) + stamp.shape[0] / 2 for k in stars}

    def batch_of(**kw):
        specs = []
        for k in range(int(g["n_src"])):
            if g["is_star"][k]:
                specs.append(amd.ComponentSpec(
                    g["sed_%d" % k], stamp, g["origin_%d" % k], sed_min_step=g["min_step_%d" % k],
                    morph_step=0.0, prox_flags=amd._lib.COMPONENT_FIXED_MORPH,
                    shift=g["center_%d" % k] - box_center[k], shift_step=3e-2))
            else:
                specs.append(amd.ComponentSpec(g["sed_%d" % k], g["morph_%d" % k], g["origin_%d" % k],
                                               sed_min_step=g["min_step_%d" % k]))
        w = np.full(g["images"].shape, 0.25, dtype=np.float32)
        return amd.BlendBatch(g["images"][None], w[None], [specs], kernel=g["diff_kernel"],
                              conv_path=path, **kw)

    batch = batch_of(max_iter=13)
    sc = point_scene(g)
    shifted = batch.model_morphologies()
    for k in stars:
        assert np.abs(shifted[k] - g["morph_%d" % k]).max() < 2e-7
    model, rendered, logL = batch.forward()
    assert rel_err(model[0], g["model"]) < RTOL
    assert rel_err(rendered[0], g["rendered"]) < RTOL
    assert abs(logL[0] - float(g["logL"])) < RTOL * abs(float(g["logL"]))
    g_sed, g_morph = batch.gradient()
    g_center = batch.centers()["gradient"]
    _, grads = sc.loss_and_gradients()
    sc.loss.clear()
    for k, c in enumerate(sc.components):
        assert np.abs(g_sed[k] - grads[k][0]).max() < 2e-5 * max(np.abs(grads[k][0]).max(), 1.0), k
        got = g_center[k] if g["is_star"][k] else g_morph[k]
        assert np.abs(got - grads[k][1]).max() < 2e-5 * np.abs(grads[k][1]).max() + 1e-3, k
    n_it = 12
    batch.step(0, n_it, e_rel=1e-4)
    for it in range(n_it):
        sc.step(it, 1e-4)
    assert_loss_close(batch.loss_history()[0], sc.loss, sc.log_norm, rtol=1e-4)
    sed, _ = batch.parameters()
    ctr = batch.centers()
    for k, c in enumerate(sc.components):
        assert rel_err(sed[k], c.sed) < 5e-4, k
        if g["is_star"][k]:
            assert np.abs(ctr["center"][k] + box_center[k] - c.center).max() < 1e-4, k
            assert_allclose(ctr["vhat"][k], c.vhat_center, rtol=2e-3)
    assert max(np.abs(sc.components[k].center - g["center_%d" % k]).max() for k in stars) > 1e-3
    batch.close()
    whole = batch_of(max_iter=100)
    n_iter, _ = whole.fit(max_iter=100, e_rel=1e-4)
    _whole_fit_against_oracle(whole.loss_history()[0], n_iter[0], point_scene(g), 1e-4)
    whole.close()


def _whole_fit_against_oracle(loss, n_iter, sc, e_rel, early=2e-5, whole=5e-4, final=RTOL):
    """same stopping iteration; chi^2 within `early` over the first twelve iterations, `whole`
    through any transient, `final` (north_star's 1e-5) for the result of the fit"""
    n_ref, _ = sc.fit(max_iter=100, e_rel=e_rel)
    assert int(n_iter) == n_ref and len(loss) == n_ref, (int(n_iter), n_ref)
    chi, ref = loss - sc.log_norm, np.array(sc.loss) - sc.log_norm
    rel = np.abs(chi - ref) / np.abs(ref)
    assert rel[:12].max() < early and rel.max() < whole and rel[-1] < final, \
        (rel[:12].max(), rel.max(), rel[-1])
    return rel


def test_config3_whole_fits_follow_the_oracle(amd):
    """BASELINE configs[2] to convergence: fit(100, e_rel=1e-4) of the 1024-blend batch, blends
    0, 511 and 1023 against ``oracle.pgm.Scene.fit`` of the same scenes -- the stopping rule
    fires in the same iteration, final chi^2 within 1e-5 (the whole-fit bar of the quickstart
    scene, now at the benchmark's own size)."""
    from oracle import pgm
    from scarlet_amd import synthetic

    n_total = 1024
    kern = synthetic.psfs()
    scenes = synthetic.make_batch(range(1234, 1234 + n_total), kernel=kern)
    comps = [[amd.ComponentSpec(s["seds"][k], s["morphs"][k], s["origins"][k],
                                sed_min_step=s["noise_rms"]) for k in range(len(s["morphs"]))]
             for s in scenes]
    batch = amd.BlendBatch(np.stack([s["data"] for s in scenes]),
                           np.stack([s["weights"] for s in scenes]), comps, kernel=kern[2],
                           max_iter=100)
    n_iter, logL = batch.fit(max_iter=100, e_rel=1e-4)
    losses = batch.loss_history()
    batch.close()
    assert len(set(n_iter.tolist())) > 10  # the blends stop in different iterations
    for i in (0, 511, 1023):
        s = scenes[i]
        sc = pgm.Scene(s["data"].shape, s["data"], s["weights"], s["diff_kernel"],
                       [pgm.Component(s["seds"][k].copy(), s["morphs"][k].copy(), s["origins"][k],
                                      sed_min_step=s["noise_rms"]) for k in range(len(s["morphs"]))])
        _whole_fit_against_oracle(losses[i], n_iter[i], sc, 1e-4)
        assert -losses[i][-1] == logL[i]


@pytest.mark.parametrize("scene", POINT_SCENES)
@pytest.mark.parametrize("path", PATHS)
def test_point_source_scene_whole_fit_follows_the_oracle(amd, path, scene):
    """BASELINE configs[3] (per-band difference kernel, three free point-source centres, two
    extended sources in 71^2 / 81^2 boxes) to convergence against the oracle: same stopping
    iteration, final chi^2 within 1e-5."""
    from conftest import point_scene

    g = golden(scene)
    batch = _point_batch(amd, g, max_iter=100, conv_path=path)
    n_iter, logL = batch.fit(max_iter=100, e_rel=1e-4)
    loss = batch.loss_history()[0]
    _whole_fit_against_oracle(loss, n_iter[0], point_scene(g), 1e-4)
    batch.close()


# ---------------------------------------------------------------- free Fourier shifts
def _shifting_batch(amd, g, hsc, **kw):
    specs = [amd.ComponentSpec(g["sed_%d" % k], g["morph_%d" % k], g["origin_%d" % k],
                               sed_min_step=g["min_step_%d" % k], shift=g["shift_%d" % k])
             for k in range(int(g["n_comp"]))]
    return amd.BlendBatch(hsc["images"][None], hsc["weights"][None], [specs],
                          kernel=hsc["diff_kernel"], **kw)


@pytest.mark.parametrize("path", PATHS)
def test_shifting_scene_forward_and_gradient(amd, hsc, path):
    """ExtendedSource(shifting=True) scene built by the reference (golden): shifted
    morphologies, model, logL against the reference's values; gradients w.r.t.
    spectra, images (pulled back through the shift) and shifts against the oracle"""
    from conftest import shifting_scene

    g = golden("hsc_shifting")
    batch = _shifting_batch(amd, g, hsc, max_iter=4, conv_path=path)
    sc = shifting_scene(g, hsc)
    shifted = batch.model_morphologies()
    _, params = batch.parameters()
    for k in range(int(g["n_comp"])):
        assert np.abs(shifted[k] - g["shifted_%d" % k]).max() < 2e-6, k
        assert_array_equal(params[k], g["morph_%d" % k].astype(np.float32))
    assert_allclose(batch.centers()["center"], [g["shift_%d" % k] for k in range(int(g["n_comp"]))])
    model, rendered, logL = batch.forward()
    assert rel_err(model[0], g["model"]) < RTOL
    assert rel_err(rendered[0], g["rendered"]) < RTOL
    assert abs(logL[0] - float(g["logL"])) < RTOL * abs(float(g["logL"]))
    g_sed, g_morph = batch.gradient()
    g_shift = batch.centers()["gradient"]
    _, grads = sc.loss_and_gradients()
    # tolerances relative to the magnitudes the float32 sums run over (sum |G||morph|;
    # the shift derivative sums |g| |d shifted / d s| ~ the image-gradient scale x pixels)
    # (2e-5: the Fourier-shifted images ring over the whole box, so every box pixel of
    # the float32 gradient image contributes its rounding to the sums)
    for k, (s_sed, s_morph) in enumerate(grad_scales(sc)):
        assert np.abs(g_sed[k] - grads[k][0]).max() < 2 * RTOL * s_sed, k
        assert np.abs(g_morph[k] - grads[k][1]).max() < 2 * RTOL * s_morph, k
        n_pix = grads[k][1].size
        assert np.abs(g_shift[k] - grads[k][2]).max() < 2 * RTOL * s_morph * np.sqrt(n_pix), k


@pytest.mark.parametrize("path", PATHS)
def test_shifting_scene_steps(amd, hsc, path):
    """10 full iterations with free shifts: losses, spectra, images, shifts, moments"""
    from conftest import shifting_scene

    g = golden("hsc_shifting")
    n_it = 10
    batch = _shifting_batch(amd, g, hsc, max_iter=n_it + 1, conv_path=path)
    sc = shifting_scene(g, hsc)
    batch.step(0, n_it, e_rel=1e-4)
    for it in range(n_it):
        sc.step(it, 1e-4)
    assert_loss_close(batch.loss_history()[0], sc.loss, sc.log_norm, rtol=2e-4)
    sed, morphs = batch.parameters()
    st = batch.centers()
    for k, c in enumerate(sc.components):
        assert rel_err(sed[k], c.sed) < 1e-3, k
        assert np.abs(morphs[k] - c.morph).max() < 2e-3, k
        assert np.abs(st["center"][k] - c.shift).max() < 2e-3, k
    assert max(np.abs(c.shift - g["shift_%d" % k]).max() for k, c in enumerate(sc.components)) > 1e-2


def test_free_shift_on_a_box_beyond_the_lds(amd):
    """A free Fourier shift (ExtendedSource(shifting=True), morphology.py:124-130, 673-676) on a
    141 x 141 box: the four image-sized work arrays of the shift kernels no longer fit the LDS
    and live in global memory.  Shifted image, gradients (image pulled back through the shift,
    the shift itself) and six iterations against the oracle."""
    from oracle import pgm

    rng = np.random.default_rng(17)
    C, H, W, n = 3, 160, 150, 141
    yy, xx = np.mgrid[:n, :n] - n // 2
    morph = (np.exp(-(yy**2 / (2 * 14.0**2) + xx**2 / (2 * 9.0**2))) +
             0.3 * np.exp(-((yy - 20) ** 2 + (xx + 12) ** 2) / (2 * 6.0**2))).astype(np.float32)
    morph /= morph.max()
    small = np.exp(-((np.mgrid[:21, :21] - 10) ** 2).sum(axis=0) / (2 * 2.5**2)).astype(np.float32)
    seds = [np.array([3.0, 2.0, 1.0], dtype=np.float32), np.array([1.0, 1.5, 2.5], dtype=np.float32)]
    origins = [(8, 4), (90, 100)]
    shifts = [np.array([0.31, -0.27]), np.array([-0.12, 0.2])]
    kernel = np.zeros((1, 9, 9), dtype=np.float32)
    g1 = np.exp(-np.arange(-4, 5) ** 2 / (2 * 1.1**2))
    kernel[0] = np.outer(g1, g1) / np.outer(g1, g1).sum()

    def scene():
        comps = [pgm.Component(seds[k].copy(), m.copy(), origins[k], sed_min_step=1e-3,
                               shift=shifts[k].copy())
                 for k, m in enumerate((morph, small))]
        return pgm.Scene((C, H, W), data, weights, kernel, comps)

    data = np.zeros((C, H, W), dtype=np.float32)
    weights = np.full((C, H, W), 4.0, dtype=np.float32)
    truth = scene()
    for c, s in zip(truth.components, ([0.6, -0.5], [0.3, 0.1])):
        c.shift[:] = s
    data = (truth.render(truth.get_model()) + rng.normal(0, 0.05, (C, H, W))).astype(np.float32)
    sc = scene()
    specs = [amd.ComponentSpec(seds[k], m, origins[k], sed_min_step=1e-3, shift=shifts[k])
             for k, m in enumerate((morph, small))]
    batch = amd.BlendBatch(data[None], weights[None], [specs], kernel=kernel, max_iter=8)
    shifted = batch.model_morphologies()
    for k, c in enumerate(sc.components):
        assert np.abs(shifted[k] - c.model_morph()).max() < 2e-6, k
    model, rendered, logL = batch.forward()
    assert rel_err(model[0], sc.get_model()) < RTOL
    g_sed, g_morph = batch.gradient()
    g_shift = batch.centers()["gradient"]
    _, grads = sc.loss_and_gradients()
    sc.loss.clear()
    for k, (s_sed, s_morph) in enumerate(grad_scales(sc)):
        assert np.abs(g_sed[k] - grads[k][0]).max() < 2 * RTOL * s_sed, k
        assert np.abs(g_morph[k] - grads[k][1]).max() < 2 * RTOL * s_morph, k
        assert np.abs(g_shift[k] - grads[k][2]).max() < 2 * RTOL * s_morph * np.sqrt(grads[k][1].size), k
    n_it = 6
    batch.step(0, n_it, e_rel=1e-4)
    for it in range(n_it):
        sc.step(it, 1e-4)
    assert_loss_close(batch.loss_history()[0], sc.loss, sc.log_norm, rtol=2e-4)
    sed, morphs = batch.parameters()
    st = batch.centers()
    for k, c in enumerate(sc.components):
        assert rel_err(sed[k], c.sed) < 1e-3, k
        assert np.abs(morphs[k] - c.morph).max() < 2e-3, k
        assert np.abs(st["center"][k] - c.shift).max() < 2e-3, k
    assert max(np.abs(c.shift - shifts[k]).max() for k, c in enumerate(sc.components)) > 1e-2
    batch.close()


def test_relative_steps_of_centres_shifts_and_the_kernel_shift(amd, hsc):
    """``relative_step`` (parameter.py:126-129: ``max(minimum, factor * X.mean())``) as the step
    rule of a point-source centre, of a free Fourier shift and of ``psf_shift``: eight
    iterations against the oracle with the same rule; the rule is in force (the parameters
    end up elsewhere than with the constant minimum)."""
    from conftest import golden, hsc_scene, point_scene, shifting_scene
    from oracle import fftconv

    n_it = 8
    # point-source centres: step = max(3e-3, 1e-3 * mean(centre in frame pixels)) ~ 3e-2
    g = golden("point_source")
    ends = []
    for rel in (1e-3, 0.0):
        specs = []
        for k in range(int(g["n_src"])):
            if g["is_star"][k]:
                specs.append(amd.PointSourceSpec(g["sed_%d" % k], g["center_%d" % k], 0.9,
                                                 sed_min_step=g["min_step_%d" % k],
                                                 center_step=3e-3, center_rel_step=rel))
            else:
                specs.append(amd.ComponentSpec(g["sed_%d" % k], g["morph_%d" % k], g["origin_%d" % k],
                                               sed_min_step=g["min_step_%d" % k]))
        w = np.full(g["images"].shape, 0.25, dtype=np.float32)
        batch = amd.BlendBatch(g["images"][None], w[None], [specs], kernel=g["diff_kernel"],
                               max_iter=n_it + 1)
        sc = point_scene(g)
        for c in sc.components:
            if hasattr(c, "center_step"):
                c.center_step, c.center_rel_step = 3e-3, rel
        batch.step(0, n_it, e_rel=1e-4)
        for it in range(n_it):
            sc.step(it, 1e-4)
        ctr = batch.centers()["center"]
        for k, c in enumerate(sc.components):
            if g["is_star"][k]:
                assert np.abs(ctr[k] - c.center).max() < 1e-4, (rel, k)
        assert_loss_close(batch.loss_history()[0], sc.loss, sc.log_norm, rtol=1e-4)
        ends.append(ctr[np.asarray(g["is_star"], bool)].copy())
        batch.close()
    assert np.abs(ends[0] - ends[1]).max() > 1e-3

    # free Fourier shifts: step = max(1e-2, 0.5 * mean(shift))
    g = golden("hsc_shifting")
    ends = []
    for rel in (0.5, 0.0):
        specs = [amd.ComponentSpec(g["sed_%d" % k], g["morph_%d" % k], g["origin_%d" % k],
                                   sed_min_step=g["min_step_%d" % k], shift=g["shift_%d" % k],
                                   shift_step=1e-2, shift_rel_step=rel)
                 for k in range(int(g["n_comp"]))]
        batch = amd.BlendBatch(hsc["images"][None], hsc["weights"][None], [specs],
                               kernel=hsc["diff_kernel"], max_iter=n_it + 1)
        sc = shifting_scene(g, hsc)
        for c in sc.components:
            c.shift_step, c.shift_rel_step = 1e-2, rel
        batch.step(0, n_it, e_rel=1e-4)
        for it in range(n_it):
            sc.step(it, 1e-4)
        st = batch.centers()["center"]
        for k, c in enumerate(sc.components):
            assert np.abs(st[k] - c.shift).max() < 2e-3, (rel, k)
        assert_loss_close(batch.loss_history()[0], sc.loss, sc.log_norm, rtol=2e-4)
        ends.append(st.copy())
        batch.close()
    assert np.abs(ends[0] - ends[1]).max() > 1e-3

    # psf_shift: step = max(1e-3, 0.2 * mean(shift))
    gp = golden("hsc_psf_shift")
    kernel = hsc["diff_kernel"]
    fft_shape = list(fftconv.fft_shape(kernel.shape, kernel.shape, padding=10, axes=(-2, -1)))
    shift0 = np.abs(gp["psf_shift"]) + 0.05
    ends = []
    for rel in (0.2, 0.0):
        sc = hsc_scene(hsc)
        for c in sc.components:
            c.source = None
        sc.psf_shift = shift0.copy()
        sc.psf_shift_step, sc.psf_shift_rel_step = 1e-3, rel
        batch = hsc_batch(amd, hsc, max_iter=n_it + 1)
        batch.set_kernel_shift(kernel, shift0, fft_shape, step=1e-3)
        batch.set_kernel_shift_relative_step(rel)
        batch.step(0, n_it, e_rel=1e-3)
        for it in range(n_it):
            sc.step(it, 1e-3)
        state = batch.kernel_shift()
        assert np.abs(state["shift"][0] - sc.psf_shift).max() < 2e-5, rel
        assert_loss_close(batch.loss_history()[0], sc.loss, sc.log_norm, rtol=5e-5)
        ends.append(state["shift"][0].copy())
        batch.close()
    assert np.abs(ends[0] - ends[1]).max() > 1e-3


# ---------------------------------------------------------------- scarlet.lite
def _lite_batch(amd, g, hsc, kind, **kw):
    from scarlet_amd import _lib

    flags = (_lib.PROX_MONOTONIC | _lib.PROX_FIT_CENTER | _lib.PROX_CENTER_ON | _lib.PROX_NORM_MAX)
    specs = []
    for k in range(int(g["n_comp"])):
        extra = dict(fista_step=float(g["fista_step"][k])) if kind == "fista" else dict(
            sed_min_step=g["noise_rms"] / 10)
        specs.append(amd.ComponentSpec(
            hsc["sed_%d" % k], hsc["morph_%d" % k], hsc["origin_%d" % k], prox_flags=flags,
            neighbor_weight="angle", min_gradient=0.0, center_floor=1e-20,
            bg_level=g["noise_rms"] * 0.25, morph_step=1e-2, **extra))
    return amd.BlendBatch(hsc["images"][None], hsc["weights"][None], [specs],
                          kernel=g["diff_kernel"], log_norm=False,
                          scheme="fista" if kind == "fista" else "amsgrad", **kw)


@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("kind", ["fista", "adaprox"])
def test_lite_loop_vs_the_reference_run(amd, hsc, kind, path):
    """scarlet.lite loop (LiteBlend.fit: lite loss, FISTA / adaprox parameters with one
    prox application, centre-fitted monotonicity, background threshold, floor 1e-20)
    against what the reference itself produced (golden): losses of the first 10
    evaluations and the state after 3 iterations"""
    g = golden("lite_" + kind)
    batch = _lite_batch(amd, g, hsc, kind, max_iter=12, conv_path=path)
    batch.step(0, 3, e_rel=1e-6, prox_max_iter=1)
    sed, morphs = batch.parameters()
    for k in range(int(g["n_comp"])):
        assert rel_err(sed[k], g["a_sed_%d" % k]) < 1e-4, k
        assert np.abs(morphs[k] - g["a_morph_%d" % k]).max() < 1e-4, k
    batch.step(3, 7, e_rel=1e-6, prox_max_iter=1)
    # lite's loss is -1/2 sum w (d - m)^2 (lite/models.py:541)
    loss = -np.array(batch.loss_history()[0])
    assert_allclose(loss, g["loss"][:10], rtol=3e-4)
    assert_allclose(loss[:4], g["loss"][:4], rtol=3e-5)


# ---------------------------------------------------------------- random configurations
@pytest.mark.parametrize("tool,args", [("fuzz_vs_oracle.py", ["24", "5"]),
                                       ("fuzz_batches.py", ["24", "5"]),
                                       ("fuzz_lite.py", ["16", "5"]),
                                       ("fuzz_facade_resize.py", ["8", "5"]),
                                       ("fuzz_seam1.py", ["60", "5"]),
                                       ("fuzz_resampler.py", ["16", "5"])])
def test_random_configurations_against_the_oracle(amd, tool, args):
    """tools/fuzz_*.py with a fixed seed: random frame / box / kernel shapes, band counts,
    weightings, sparsity, point sources, shifts, ragged batches, sub-ranges, lite loops,
    facade fits with box resizing, the four seam-1 operators -- forward, gradients and a
    few iterations against the oracle.  (This is the harness that
    found the rocFFT transposed-shape defect.)"""
    import subprocess
    import sys

    from conftest import ROOT

    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool)] + args,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "OVER" not in out.stdout, out.stdout[-3000:]
    assert any(word in out.stdout for word in ("worst", "mismatches: 0")), out.stdout[-2000:]


def test_symmetry_strength(amd):
    """SymmetryConstraint(strength) (constraint.py:262-273; operator.py:274-293): the
    reference's known answers for strength 1 and 0.5 (tests/test_constraint.py:137-161)
    through the facade class, and the constraint inside the device chain for even and odd
    boxes against the oracle."""
    from scarlet_amd import _lib
    from scarlet_amd.constraint import SymmetryConstraint, device_flags

    x = np.arange(25, dtype=float).reshape(5, 5)
    assert_allclose(SymmetryConstraint()(x.copy(), 0), np.full((5, 5), 12.0))
    assert_allclose(SymmetryConstraint(strength=0.5)(x.copy(), 0), x * 0.5 + 6.0)
    assert device_flags(SymmetryConstraint(0.5))["sym_strength"] == 0.5
    rng = np.random.default_rng(17)
    boxes = [((21, 21), (3, 5)), ((20, 31), (20, 25)), ((25, 34), (30, -4))]
    specs, kernel, data, weights = _random_scene(rng, 3, 64, 72, boxes, kernel_shape=15)
    flags = _lib.PROX_EXTENDED_SOURCE | _lib.PROX_SYMMETRY
    for strength in (0.5, 0.2):
        _compare_steps(amd, specs, kernel, data, weights, 3, dict(sym_strength=strength),
                       dict(symmetric=strength), flags=flags)


def test_constraint_chain_repeat(amd):
    """ConstraintChain(*constraints, repeat=n) (constraint.py:60-80): the chain applied n
    times per proximal evaluation -- through the facade class on the host, and inside the
    device loop (general update kernel) against the oracle."""
    from scarlet_amd import _lib
    from scarlet_amd.constraint import (ConstraintChain, NormalizationConstraint,
                                        PositivityConstraint, SymmetryConstraint, device_flags)

    chain = ConstraintChain(SymmetryConstraint(0.5), PositivityConstraint(),
                            NormalizationConstraint("max"), repeat=3)
    x = np.arange(25, dtype=float).reshape(5, 5) - 3
    once = ConstraintChain(*chain.constraints)
    assert_allclose(chain(x.copy(), 0), once(once(once(x.copy(), 0), 0), 0))
    assert device_flags(chain)["chain_repeat"] == 3
    rng = np.random.default_rng(23)
    boxes = [((21, 21), (3, 5)), ((20, 31), (20, 25)), ((25, 34), (30, -4))]
    specs, kernel, data, weights = _random_scene(rng, 3, 64, 72, boxes, kernel_shape=15)
    flags = _lib.PROX_EXTENDED_SOURCE | _lib.PROX_SYMMETRY
    batch, sc = _compare_steps(amd, specs, kernel, data, weights, 3,
                               dict(sym_strength=0.5, chain_repeat=3),
                               dict(symmetric=0.5), flags=flags,
                               oracle_attrs=dict(chain_repeat=3))
    batch.close()
    # PositivityConstraint(zero=0.02) on the morphology (constraint.py:83-92), in both
    # update kernels (the repeating chain takes the general one)
    assert device_flags(ConstraintChain(PositivityConstraint(zero=0.02)))["zero"] == 0.02
    for repeat in (1, 2):
        batch, sc = _compare_steps(amd, specs, kernel, data, weights, 3,
                                   dict(pos_floor=0.02, chain_repeat=repeat), dict(),
                                   oracle_attrs=dict(morph_zero=0.02, chain_repeat=repeat))
        # floored before the normalisation by the maximum: no pixel is left at zero
        assert min(float(m.min()) for m in batch.parameters()[1]) > 0.005
        batch.close()


def _step_bits(amd, data, weights, comps, kernel, n_it, inline, sub_ranges=1):
    batch = amd.BlendBatch(data, weights, comps, kernel=kernel, max_iter=n_it + 1, conv_path="fused")
    batch.set_inline_render(inline)
    batch.set_sub_ranges(sub_ranges)
    batch.step(0, n_it, e_rel=1e-3)
    sed, morphs = batch.parameters()
    out = (np.concatenate(batch.loss_history()), sed.copy(),
           np.concatenate([m.ravel() for m in morphs]))
    batch.close()
    return out


@pytest.mark.parametrize("case", ["benchmark boxes", "wide boxes", "overhanging boxes",
                                  "70 components", "too wide for the kernel"])
def test_rows_rendered_inside_the_convolution_equal_the_model_cube(amd, case):
    """A plain batch has no model cube: the convolution kernel renders its input rows itself
    (Blend.get_model, blend.py:200-244, in the stride-pass layout of the row transforms).
    The rows must be bit for bit what render_kernel writes -- same terms, same order, one fma
    each -- so every loss and every parameter after three full iterations is identical to
    the run that keeps the cube: 41^2 boxes (four columns of a residue class per box), 61^2 /
    81^2 boxes (six), boxes that overhang all four edges of a frame whose sides are no
    multiples of 16 or 2, more components than a wavefront has lanes, and a 101-pixel box,
    which sends the batch back to the cube (nothing to compare but that it still runs)."""
    rng = np.random.default_rng(len(case))
    C, H, W, ks = 3, 128, 128, 15
    if case == "benchmark boxes":
        boxes = [((41, 41), (int(rng.integers(0, 88)), int(rng.integers(0, 88)))) for _ in range(10)]
    elif case == "wide boxes":
        boxes = [((61, 61), (3, 40)), ((81, 81), (30, 15)), ((21, 71), (100, 50)), ((41, 41), (60, 80))]
    elif case == "overhanging boxes":
        H, W = 101, 119
        boxes = [((41, 41), (-20, -17)), ((31, 45), (85, 100)), ((25, 25), (-5, 100)),
                 ((35, 21), (80, -10)), ((41, 41), (30, 37)), ((15, 61), (50, 70))]
    elif case == "70 components":
        H, W = 56, 66
        boxes = [((9 + 2 * (k % 3), 9 + 2 * (k % 4)), (int(rng.integers(-3, 50)), int(rng.integers(-3, 60))))
                 for k in range(70)]
    else:
        boxes = [((101, 101), (10, 12)), ((41, 41), (60, 80))]
    scenes = [_random_scene(rng, C, H, W, boxes, kernel_shape=ks) for _ in range(3)]
    kernel = scenes[0][1]
    data = np.stack([s[2] for s in scenes])
    weights = np.stack([s[3] for s in scenes])
    comps = [[amd.ComponentSpec(sed * 0.8, m, o, sed_min_step=0.05) for sed, m, o in s[0]] for s in scenes]
    ref = _step_bits(amd, data, weights, comps, kernel, 3, inline=False)
    for sub in (1, 3):
        got = _step_bits(amd, data, weights, comps, kernel, 3, inline=True, sub_ranges=sub)
        for a, b in zip(got, ref):
            assert np.all(np.isfinite(a))
            assert_array_equal(a, b)
