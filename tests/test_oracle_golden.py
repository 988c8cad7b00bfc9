"""The CPU oracle against (a) the reference's own literal known-answer tables and
(b) golden vectors produced by running the reference (oracle/refshim/make_golden.py).
CPU only."""

import numpy as np
import pytest
from numpy.testing import assert_allclose, assert_array_equal, assert_almost_equal

from conftest import golden, hsc_scene
from oracle import fftconv, proxops, pgm


# ---- G1: centring tables, transcribed from the reference's tests/test_fft.py:12-80
def test_pad_and_center_tables():
    a_pad = fftconv.pad_to(np.ones((1, 1)), (5, 4))
    truth = np.zeros((5, 4))
    truth[2, 2] = 1
    assert_array_equal(a_pad, truth)
    shifted = np.fft.ifftshift(a_pad)
    t2 = np.zeros((5, 4))
    t2[0, 0] = 1
    assert_array_equal(shifted, t2)

    a0 = np.arange(10).reshape(5, 2)
    a_pad = fftconv.pad_to(a0, (9, 11))
    truth = np.zeros((9, 11), dtype=int)
    truth[2:7, 5:7] = a0
    assert_array_equal(a_pad, truth)
    shifted = np.fft.ifftshift(a_pad)
    t2 = np.zeros((9, 11), dtype=int)
    t2[0:3, 0:2] = a0[2:5]
    t2[7:9, 0:2] = a0[0:2]
    assert_array_equal(shifted, t2)
    assert_array_equal(np.fft.fftshift(shifted), a_pad)
    assert_array_equal(fftconv.centered(a_pad, (5, 2)), a0)


# ---- G4: monotonic 5x5 tables, reference tests/test_constraint.py:92-135
MONO_NEAREST = [
    [0.0, 1.0, 2.0, 3.0, 4.0],
    [5.0, 6.0, 7.0, 8.0, 9.0],
    [10.0, 11.0, 12.0, 12.0, 12.0],
    [11.0, 12.0, 12.0, 12.0, 12.0],
    [12.0, 12.0, 12.0, 12.0, 12.0],
]
MONO_ANGLE = [
    [0.000000000, 1.000000000, 2.000000000, 3.000000000, 4.000000000],
    [5.000000000, 6.000000000, 7.000000000, 8.000000000, 9.000000000],
    [9.742640687, 11.000000000, 12.000000000, 12.000000000, 10.828427125],
    [11.030627697, 11.707106781, 12.000000000, 12.000000000, 11.771236166],
    [11.556349186, 11.868867239, 11.914213562, 11.983249156, 11.928090416],
]
MONO_ANGLE_G25 = [
    [0.000000000, 1.000000000, 2.000000000, 3.000000000, 4.000000000],
    [5.000000000, 6.000000000, 7.000000000, 7.242640687, 5.806841831],
    [5.801461031, 9.000000000, 12.000000000, 9.000000000, 6.074431804],
    [5.895545844, 7.681980515, 9.000000000, 7.681980515, 5.935521488],
    [4.988519641, 5.949655012, 6.170941546, 5.949655012, 4.997301087],
]


@pytest.mark.parametrize(
    "mode,g,truth",
    [("nearest", 0, MONO_NEAREST), ("angle", 0, MONO_ANGLE), ("angle", 0.25, MONO_ANGLE_G25)],
)
def test_monotonic_known_answers(mode, g, truth):
    x = np.arange(25, dtype=float).reshape(5, 5)
    out = proxops.prox_monotonic(x.copy(), 0, mode, g)
    assert_almost_equal(out, truth)
    # the pure-Python statement of the loop agrees bit for bit with the C one
    w, didx, off = proxops.monotonic_operator((5, 5), mode, (2, 2))
    assert_array_equal(proxops.sweep_py(x.copy(), w, off, didx, g), out)


# ---- G5: symmetry tables and threshold constant, tests/test_constraint.py:73-90,137-161
def test_symmetry_and_threshold_known_answers():
    x = np.arange(25, dtype=float).reshape(5, 5)
    assert_almost_equal(proxops.prox_soft_symmetry(x.copy(), 0, 1), np.full((5, 5), 12.0))
    half = np.arange(25, dtype=float).reshape(5, 5) * 0.5 + 6.0
    assert_almost_equal(proxops.prox_soft_symmetry(x.copy(), 0, 0.5), half)

    np.random.seed(0)
    noise = np.random.rand(21, 21) * 2
    psf = golden_gaussian(1.0, 21)
    signal = np.zeros((21, 21))
    signal[7:14, 7:14] = psf[7:14, 7:14]
    X = signal + noise
    out = proxops.prox_threshold(X.copy())
    # the reference's regression assertion, with its stored constant
    thresh = 0.05704869232578929
    mask = X < thresh
    assert np.all(out[mask] == 0)
    assert_array_equal(out[~mask], X[~mask])


def golden_gaussian(sigma, boxsize):
    """Pixel-integrated Gaussian as psf.py:128-142, via the product's PSF class
    (host set-up code), checked against the golden PSFs below."""
    from scarlet_amd.psf import GaussianPSF

    return GaussianPSF(sigma, boxsize=boxsize).get_model()[0]


def test_prox_soft_hard_semantics():
    # tests/test_constraint.py:35-71
    rng = np.random.default_rng(0)
    X = rng.random(100) - 0.5
    step, thresh = 0.5, 0.25
    for typ, t in (("relative", thresh * step), ("absolute", thresh)):
        out = proxops.prox_hard(X.copy(), step, thresh=thresh, type=typ)
        mask = np.abs(X) < t
        assert np.all(out[mask] == 0)
        assert_array_equal(out[~mask], X[~mask])
        out = proxops.prox_soft(X.copy(), step, thresh=thresh, type=typ)
        assert np.all(out[mask] == 0)
        assert_array_equal(np.abs(out[~mask]), np.abs(np.abs(X[~mask]) - t))


# ---- G10: operator set-up tables from the reference
def test_operator_tables_match_reference():
    g = golden("operator_tables")
    for key in g.files:
        if key.startswith("didx_"):
            h, w = map(int, key[5:].split("x"))
            assert_array_equal(proxops.sort_by_radius((h, w), (h // 2, w // 2)), g[key])
        elif key.startswith("w_"):
            _, mode, tag = key.split("_")
            h, w = map(int, tag.split("x"))
            mine = proxops.radial_monotonic_weights((h, w), mode, (h // 2, w // 2))
            assert_allclose(mine, g[key], rtol=0, atol=1e-15)
    for tag in ("21x21", "31x41"):
        x0 = g["sweep_in_" + tag]
        for mode, gr in (("flat", 0.1), ("angle", 0.0), ("nearest", 0.0), ("angle", 0.25)):
            out = proxops.prox_monotonic(x0.copy(), 0, mode, gr)
            assert_array_equal(out, g["sweep_{}_{}_{}".format(mode, gr, tag)])


def _offsets(w):
    """operator.getOffsets (reference operator.py:512-527)"""
    return np.array([-w - 1, -w, -w + 1, -1, 1, w - 1, w, w + 1], dtype=np.int32)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("tag", ["7x9", "21x21"])
def test_sweep_against_a_transcription_of_the_reference_loop(dtype, tag):
    """The sweep goldens of ``operator_tables.npz`` went through oracle/csrc/sweep.c when
    they were made (the reference's compiled module could not be built, its Python calls
    were routed to the oracle).  This test pins the sweep without that file: the C++ loop
    (operators_pybind11.cc:14-36) transcribed line by line into scalar Python
    (tests/mask_kats.py::sweep_transcription), run on the weight / order tables that the
    reference's own pure-NumPy set-up code produced (didx_*, w_* in the golden), must give
    bit for bit what the C restatement gives."""
    from mask_kats import sweep_transcription

    g = golden("operator_tables")
    h, w = map(int, tag.split("x"))
    rng = np.random.default_rng(h * 100 + w)
    didx = g["didx_" + tag][1:]  # the peak itself is not swept (operator.py:92)
    for mode, gr in (("flat", 0.1), ("angle", 0.0), ("nearest", 0.0), ("angle", 0.25)):
        wts = g["w_{}_{}".format(mode, tag)]
        x0 = (rng.random((h, w)) + 0.1 * rng.random((h, w)) * (mode == "flat")).astype(dtype)
        want = sweep_transcription(x0.copy(), wts, _offsets(w), didx, gr)
        got = proxops.sweep(x0.copy(), wts, _offsets(w), didx, gr)
        assert_array_equal(got, want)
        assert not np.array_equal(want, x0)  # the sweep clipped something


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_mask_operators_hand_derived_known_answers(dtype):
    """oracle/csrc/mask.c against the known answers derived by hand from
    operators_pybind11.cc:61-232 (tests/mask_kats.py)"""
    import mask_kats

    for name, img, (i, j), var, thr, unc_w, orp_w, b_w in mask_kats.valid_pixel_cases(dtype):
        unchecked = np.ones(img.shape, dtype=bool)
        unchecked[i, j] = False
        orphans = np.zeros(img.shape, dtype=bool)
        bounds = np.array([i, i, j, j], dtype=np.int32)
        proxops.get_valid_monotonic_pixels(i, j, img, unchecked, orphans, var, bounds, thr)
        assert_array_equal(unchecked, unc_w, err_msg=name)
        assert_array_equal(orphans, orp_w, err_msg=name)
        assert bounds.tolist() == b_w, name
    for name, model, unc, orp, oi, oj, rec, b0, m_w, unc_w, orp_w, b_w in \
            mask_kats.interpolation_cases(dtype):
        model, unc, orp = model.copy(), unc.copy(), orp.copy()
        bounds = np.array(b0, dtype=np.int32)
        proxops.linear_interpolate_invalid_pixels(oi, oj, unc, model, orp, 0.0, rec, bounds)
        assert_array_equal(model, m_w, err_msg=name)
        assert_array_equal(unc, unc_w, err_msg=name)
        assert_array_equal(orp, orp_w, err_msg=name)
        assert bounds.tolist() == b_w, name


# ---- G2: PSF matching and convolution
def test_fft_psf_matching_golden():
    g = golden("fft_psf")
    k12 = fftconv.match_psf(g["psf2"], g["psf1"])
    assert_allclose(k12, g["k12"], rtol=0, atol=1e-12)
    assert_allclose(fftconv.convolve(g["psf1"], k12), g["img2"], rtol=0, atol=1e-12)
    assert_almost_equal(g["img2"], g["psf2"])  # the reference's own identity
    assert_allclose(fftconv.match_psf(g["psf1"], g["psf2"]), g["k21"], rtol=1e-9, atol=1e-9)
    km = fftconv.match_psf(g["psf123"], g["psf1"])
    assert_allclose(km, g["kmulti"], rtol=0, atol=1e-12)
    # the reference reuses the cached k-space ratio of `kmulti` here -> 1e-10
    assert_allclose(fftconv.convolve(km, g["psf1"]), g["imulti"], rtol=0, atol=1e-10)
    conv = fftconv.convolve(g["cube"], g["kern"], axes=(1, 2))
    assert conv.dtype == np.float32
    assert_allclose(conv, g["conv"], rtol=0, atol=1e-6)
    shapes = [
        fftconv.fft_shape((5, 58, 48), (5, 43, 43), 3, (1, 2)),
        fftconv.fft_shape((5, 128, 128), (1, 41, 41), 3, (1, 2)),
        fftconv.fft_shape((6, 40, 59), (6, 31, 31), 3, (1, 2)),
        fftconv.fft_shape((1, 43, 43), (1, 9, 9), 10, (-2, -1)),
        fftconv.fft_shape((2, 30, 30), (2, 10, 12), 3, (1, 2)),
    ]
    assert_array_equal(np.array(shapes), g["fft_shapes"])
    assert_allclose(
        fftconv.fourier_shift(g["shift_in"], (0.3, -1.7)), g["shift_out"], rtol=0, atol=1e-6
    )


def test_fft_equals_direct_convolution_and_adjoint():
    rng = np.random.default_rng(2)
    img = rng.standard_normal((2, 30, 37))
    ker = rng.standard_normal((2, 9, 7))
    fftc = fftconv.convolve(img, ker, axes=(1, 2))
    for c in range(2):
        assert_allclose(fftc[c], fftconv.apply_filter(img[c], ker[c]), atol=1e-11)
    y = rng.standard_normal(img.shape)
    lhs = np.sum(fftc * y)
    rhs = np.sum(img * fftconv.convolve_adjoint(y, ker, axes=(1, 2)))
    assert_allclose(lhs, rhs, rtol=1e-12)
    flipped = fftconv.convolve(y, ker[:, ::-1, ::-1], axes=(1, 2))
    assert_allclose(fftconv.convolve_adjoint(y, ker, axes=(1, 2)), flipped, atol=1e-11)


# ---- G3: render + loss (reference tests/test_observation.py:13-47)
def test_render_loss_golden():
    g = golden("render_loss")
    # renderer.py:197-202 casts both PSFs to the frame dtype (float32) first
    kernel = fftconv.match_psf(
        g["obs_psf"].astype(np.float32), g["model_psf"].astype(np.float32), padding=10
    )
    assert_allclose(kernel, g["diff_kernel"], rtol=0, atol=1e-7)
    sc = pgm.Scene(g["model"].shape, g["images"], np.ones_like(g["images"]), kernel, [],
                   dtype=np.float64)
    rendered = sc.render(g["model"])
    assert_allclose(rendered, g["rendered"], rtol=0, atol=1e-7)
    assert_almost_equal(rendered, g["obs_psf"])  # the reference's assertion
    assert_allclose(sc.log_norm, g["log_norm"], rtol=1e-14)
    assert_allclose(sc.log_likelihood(rendered), g["logL"], rtol=1e-9)


# ---- G6/G8: the quickstart scene
def test_hsc_forward_golden(hsc):
    g = hsc
    sc = hsc_scene(g)
    kernel = fftconv.match_psf(
        g["psfs"].astype(np.float32), g["model_psf"].astype(np.float32), padding=10
    )
    assert kernel.dtype == np.float32
    assert_allclose(kernel, g["diff_kernel"], rtol=0, atol=1e-7)
    model = sc.get_model()
    assert model.dtype == np.float32
    assert_array_equal(model, g["model"])
    rendered = sc.render(model)
    assert_allclose(rendered, g["rendered"], rtol=0, atol=1e-5 * np.abs(g["rendered"]).max())
    assert_allclose(sc.log_norm, g["log_norm"], rtol=1e-6)
    assert_allclose(sc.log_likelihood(rendered), g["logL"], rtol=1e-6)
    assert_allclose(float(g["logL"]), -357918.244, atol=0.05)


def test_hsc_gradient_matches_finite_differences_of_the_reference(hsc):
    g = hsc
    sc = hsc_scene(g, dtype64=True)
    _, grads = sc.loss_and_gradients()
    n = int(g["n_comp"])
    for j in range(len(g["fd_dlogL"])):
        dot = 0.0
        for k in range(n):
            dot += np.sum(grads[k][0] * g["dir%d_%d" % (j, 2 * k)].astype(np.float64))
            dot += np.sum(grads[k][1] * g["dir%d_%d" % (j, 2 * k + 1)].astype(np.float64))
        # fd is d(logL); our gradients are of the loss -logL
        assert_allclose(-dot, g["fd_dlogL"][j], rtol=2e-6)


def test_psf_unmatched_golden():
    g = golden("psf_unmatched")
    kernel = fftconv.match_psf(g["psfs"], g["model_psf"].astype(np.float32), padding=10)
    assert_allclose(kernel, g["diff_kernel"], rtol=0, atol=1e-6)
    w = np.ones_like(g["images"]) / 4
    sc = pgm.Scene(g["images"].shape, g["images"], w, g["diff_kernel"], [])
    rendered = sc.render(g["model"])
    assert_allclose(rendered, g["rendered"], rtol=0, atol=2e-6)
    assert_allclose(sc.log_likelihood(rendered), g["logL"], rtol=1e-6)


@pytest.mark.parametrize("name", ["point_source", "point_source_moffat", "point_source_image",
                                  "point_source_bands"])
def test_point_source_scene_golden(name):
    """docs/tutorials/point_source.ipynb scene built by the reference (on its GaussianPSF
    model PSF, on a MoffatPSF, on an ImagePSF and on an ImagePSF that differs between the
    bands -- the morphology is a cube then): PSF morphology at a sub-pixel centre (bit-exact), model,
    rendered image, logL, and the gradient (centres included) against finite differences of
    the reference's forward."""
    from conftest import point_scene

    g = golden(name)
    sc = point_scene(g)
    for k, c in enumerate(sc.components):
        assert tuple(c.origin) == tuple(g["origin_%d" % k])
        if g["is_star"][k]:
            assert_array_equal(c.morph, g["morph_%d" % k])
    model = sc.get_model()
    assert_array_equal(model, g["model"])
    rendered = sc.render(model)
    assert_allclose(rendered, g["rendered"], rtol=0, atol=1e-5 * np.abs(g["rendered"]).max())
    assert_allclose(sc.log_likelihood(rendered), g["logL"], rtol=1e-6)

    sc64 = point_scene(g, dtype64=True)
    _, grads = sc64.loss_and_gradients()
    for j in range(len(g["fd_dlogL"])):
        dot = 0.0
        for k in range(int(g["n_src"])):
            dot += np.sum(grads[k][0] * g["dir%d_%d" % (j, 2 * k)])
            dot += np.sum(grads[k][1] * g["dir%d_%d" % (j, 2 * k + 1)])
        assert_allclose(-dot, g["fd_dlogL"][j], rtol=2e-6)


def test_shifting_scene_golden(hsc):
    """ExtendedSource(shifting=True) built by the reference: Fourier-shifted
    morphologies, model, logL; the separable shift operator equals the FFT
    implementation; gradient (shifts included) against finite differences of the
    reference's forward."""
    from conftest import shifting_scene

    g = golden("hsc_shifting")
    sc = shifting_scene(g, hsc)
    for k, c in enumerate(sc.components):
        assert np.abs(c.shift).max() > 0
        shifted = c.model_morph()
        assert_allclose(shifted, g["shifted_%d" % k], rtol=0, atol=1e-14)
        op = fftconv.ShiftOperator(c.morph.shape, c.shift)
        assert_allclose(op.forward(c.morph), shifted, rtol=0, atol=1e-13)
    model = sc.get_model()
    assert_allclose(model, g["model"], rtol=0, atol=1e-6 * np.abs(g["model"]).max())
    rendered = sc.render(model)
    assert_allclose(sc.log_likelihood(rendered), g["logL"], rtol=1e-6)

    sc64 = shifting_scene(g, hsc, dtype64=True)
    _, grads = sc64.loss_and_gradients()
    for j in range(len(g["fd_dlogL"])):
        dot = 0.0
        for k in range(int(g["n_comp"])):
            for i in range(3):
                dot += np.sum(grads[k][i] * g["dir%d_%d" % (j, 3 * k + i)].astype(np.float64))
        assert_allclose(-dot, g["fd_dlogL"][j], rtol=2e-6)


@pytest.mark.parametrize("kind", ["fista", "adaprox"])
def test_lite_fit_matches_the_reference_run(hsc, kind):
    """LiteBlend.fit run by the reference in the build container (FISTA: every line of
    the loop is reference code; adaprox: the reference's loop around the shim's AMSGrad
    moments): losses of all 26 evaluations, state after 3 iterations, boxes and state
    after 25 iterations with two resize rounds."""
    from conftest import lite_scene

    g = golden("lite_" + kind)
    sc = lite_scene(g, hsc, kind)
    assert_allclose(sc.kernel, g["diff_kernel"])
    it, _ = sc.fit(3, e_rel=1e-9, resize=10)
    assert it == int(g["it_a"])
    # same NumPy, same arithmetic: the restatement reproduces the reference bit for bit
    for k, c in enumerate(sc.components):
        assert_array_equal(c.sed, g["a_sed_%d" % k])
        assert_array_equal(c.morph, g["a_morph_%d" % k])
    it, _ = sc.fit(25, e_rel=1e-9, resize=10)
    assert it == int(g["it_b"]) and len(sc.loss) == len(g["loss"])
    assert_array_equal(np.array(sc.loss), g["loss"])
    for k, c in enumerate(sc.components):
        assert c.morph.shape == g["b_morph_%d" % k].shape, k
        assert tuple(c.origin) == tuple(g["b_origin_%d" % k]), k
        assert_array_equal(c.sed, g["b_sed_%d" % k])
        assert_array_equal(c.morph, g["b_morph_%d" % k])


def test_psf_shift_golden(hsc):
    """ConvolutionRenderer(psf_shift=...) run by the reference: rendered cube and logL at
    a non-zero kernel shift, and the gradient w.r.t. the shift against finite differences
    of the reference's forward"""
    gp = golden("hsc_psf_shift")
    sc = hsc_scene(hsc)
    sc.psf_shift = gp["psf_shift"].copy()
    rendered = sc.render(gp["model"])
    assert_allclose(rendered, gp["rendered"], rtol=0, atol=1e-5 * np.abs(gp["rendered"]).max())
    assert_allclose(sc.log_likelihood(rendered), gp["logL"], rtol=1e-6)
    sc64 = hsc_scene(hsc, dtype64=True)
    sc64.psf_shift = gp["psf_shift"].copy()
    g = sc64.psf_shift_gradient(gp["model64"], sc64.render(gp["model64"]))
    assert_allclose(-g, gp["fd_dlogL_dshift"], rtol=1e-6)


def test_synthetic_cfg2_golden():
    from scarlet_amd import synthetic

    g = golden("synthetic_cfg2")
    s = synthetic.make_blend(1234)
    assert_allclose(s["data"].astype(np.float64).sum(), g["data_checksum"], rtol=1e-12)
    assert_allclose(s["diff_kernel"], g["diff_kernel"], rtol=0, atol=1e-7)
    comps = [pgm.Component(s["seds"][k].copy(), s["morphs"][k].copy(), s["origins"][k])
             for k in range(len(s["morphs"]))]
    sc = pgm.Scene(s["data"].shape, s["data"], s["weights"], s["diff_kernel"], comps)
    model = sc.get_model()
    assert_array_equal(model, g["model"])
    rendered = sc.render(model)
    assert_allclose(rendered, g["rendered"], rtol=0, atol=1e-5 * np.abs(g["rendered"]).max())
    # logL = -log_norm - chi2/2 is a difference of large numbers: compare chi2/2
    chi2 = -(sc.log_likelihood(rendered) + sc.log_norm)
    assert_allclose(chi2, -(g["logL"] + g["log_norm"]), rtol=1e-6)
    assert_allclose(sc.log_norm, g["log_norm"], rtol=1e-7)


# ---- optimizer restatement: self-consistency (parity unpinned, see oracle/__init__.py)
def test_adaprox_update_properties():
    rng = np.random.default_rng(4)
    x = rng.random(7) + 1.0
    g = rng.standard_normal(7)
    m, v, vh = np.zeros(7), np.zeros(7), np.zeros(7)
    x0 = x.copy()
    pgm.adaprox_update(0, x, g, m, v, vh, 0.1, None, 1e-3)
    assert_allclose(m, 0.1 * g)
    assert_allclose(v, 0.001 * g * g)
    assert_array_equal(vh, v)  # vhat = v on the first iteration
    psi = np.sqrt(np.maximum(vh, 1e-8))
    assert_allclose(x, x0 - 0.1 * m / psi / 10)  # a tenth of the step at it = 0
    v_prev = vh.copy()
    pgm.adaprox_update(1, x, g * 0.01, m, v, vh, 0.1, None, 1e-3)
    assert np.all(vh >= v_prev)  # amsgrad: vhat never decreases


def test_fit_increases_likelihood(hsc):
    sc = hsc_scene(hsc)
    n, logL = sc.fit(max_iter=15, e_rel=1e-4)
    assert n == len(sc.loss) == 15
    assert -sc.loss[-1] > -sc.loss[0]
    for c in sc.components:
        assert c.morph.max() == 1.0 and c.morph.min() >= 0 and np.all(c.sed > 0)


def test_multiresolution_fit_term_golden():
    """Two observations, one on a coarser grid (ResolutionRenderer): the oracle's
    restatement of the per-call path (oracle/resample.py) against the reference's own
    float64 evaluation -- model, both renderings, both log-likelihoods -- and its analytic
    gradient against central finite differences of the reference's total -logL."""
    from multires_scene import build

    g = golden("multires_fit")
    scene, lowres, c_hr, ((dy0, dy1), (dx0, dx1), (my0, my1), (mx0, mx1)) = build(g)
    model = scene.get_model()
    assert_allclose(model, g["model"], rtol=0, atol=1e-12 * np.abs(g["model"]).max())
    peak_lr = np.abs(g["rendered_lr"]).max()
    assert_allclose(lowres.render(model), g["rendered_lr"], rtol=0, atol=1e-12 * peak_lr)
    rendered = scene.render(model)
    assert_allclose(rendered[c_hr, my0:my1, mx0:mx1], g["rendered_hr"][0, dy0:dy1, dx0:dx1],
                    rtol=0, atol=1e-12 * np.abs(g["rendered_hr"]).max())
    assert_allclose(lowres.log_norm, g["log_norm_lr"], rtol=1e-13)
    term, _ = lowres.neg_log_likelihood(model)
    assert_allclose(-term, g["logL_lr"], rtol=1e-12)
    loss, grads = scene.loss_and_gradients()
    assert_allclose(-loss, g["logL_hr"] + g["logL_lr"], rtol=1e-12)
    # adjoint identity <R m, u> = <m, R^T u>
    rng = np.random.default_rng(5)
    m = rng.normal(size=model.shape)
    u = rng.normal(size=g["rendered_lr"].shape)
    assert_allclose(np.sum(lowres.render(m) * u), np.sum(m * lowres.adjoint(u, model.shape[0])),
                    rtol=1e-11)
    for k, (g_sed, g_morph) in enumerate(grads):
        fd_sed, fd_morph = g["fd_%d" % (3 * k)], g["fd_%d" % (3 * k + 1)]
        assert np.abs(g_sed - fd_sed).max() < 1e-7 * np.abs(fd_sed).max()
        assert np.abs(g_morph - fd_morph).max() < 1e-6 * np.abs(fd_morph).max()


def test_float32_state_mode_of_the_oracle(hsc):
    """``state_dtype=np.float32`` (oracle/pgm.py: the device's precision -- float32 moments
    and optimizer arithmetic) against the reference-faithful float64 state on the whole
    quickstart fit: same iteration count, same result to 1e-6, and a transient difference
    of ~3e-4 around iteration 25 that is a property of the scene, not of either
    implementation (the GPU tests bound the device against both modes)."""
    import numpy as np
    from conftest import hsc_scene

    runs = {}
    for dt in (np.float64, np.float32):
        sc = hsc_scene(hsc, state_dtype=dt)
        n, _ = sc.fit(max_iter=100, e_rel=1e-4)
        assert sc.components[0].m_morph.dtype == dt and sc.components[0].v_sed.dtype == dt
        runs[dt] = (n, np.array(sc.loss) - sc.log_norm)
    (n64, a), (n32, b) = runs[np.float64], runs[np.float32]
    assert n64 == n32 == 76
    rel = np.abs(a - b) / np.abs(a)
    assert rel[:12].max() < 2e-5 and rel.max() < 1e-3 and rel[-1] < 1e-6
    assert rel.max() > 5e-5  # the transient exists on the CPU alone
