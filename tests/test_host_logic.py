"""Host-side mirror of the reference interface (no GPU): boxes, FFT utilities,
PSFs, monotonicity tables, constraints and their translation to the device chain.
The expectations are the reference's own tests (tests/test_bbox.py, test_fft.py,
test_component.py, test_constraint.py) and the golden vectors."""

import pickle

import numpy as np
import pytest
from numpy.testing import assert_allclose, assert_almost_equal, assert_array_equal

import scarlet_amd as scarlet
from scarlet_amd import _lib, fft, operator
from scarlet_amd.constraint import device_flags
from conftest import golden
from oracle import fftconv, proxops


def test_box_algebra():
    # reference tests/test_bbox.py
    box = scarlet.Box((3, 4), origin=(1, 2))
    assert box.D == 2 and box.start == (1, 2) and box.stop == (4, 6)
    assert box.bounds == ((1, 4), (2, 6)) and box.slices == (slice(1, 4), slice(2, 6))
    assert scarlet.Box.from_bounds((1, 4), (2, 6)) == box
    assert (box | scarlet.Box((2, 2), origin=(0, 0))) == scarlet.Box((4, 6), origin=(0, 0))
    assert (box & scarlet.Box((3, 3), origin=(2, 3))) == scarlet.Box((2, 3), origin=(2, 3))
    assert (box & scarlet.Box((1, 1), origin=(9, 9))).shape == (0, 0)
    assert box + 2 == scarlet.Box((3, 4), origin=(3, 4))
    assert box - (1, 2) == scarlet.Box((3, 4))
    assert scarlet.Box((5,)) @ box == scarlet.Box((5, 3, 4), origin=(0, 1, 2))
    assert box.contains((1, 2)) and not box.contains((4, 2))
    img = np.arange(30).reshape(5, 6)
    sub = scarlet.Box((3, 3), origin=(-1, 4)).extract_from(img)
    assert_array_equal(sub, [[0, 0, 0], [4, 5, 0], [10, 11, 0]])
    x = np.zeros((5, 6))
    x[2:, 3:] = 1
    assert scarlet.Box.from_data(x) == scarlet.Box((3, 3), origin=(2, 3))
    f_sl, b_sl = scarlet.overlapped_slices(scarlet.Box((5, 6)), scarlet.Box((3, 3), origin=(-1, 4)))
    assert f_sl == (slice(0, 2), slice(4, 6)) and b_sl == (slice(1, 3), slice(0, 2))


def test_fft_centering_conventions():
    # reference tests/test_fft.py:12-80
    a_pad = fft._pad(np.ones((1, 1)), (5, 4))
    truth = np.zeros((5, 4))
    truth[2, 2] = 1
    assert_array_equal(a_pad, truth)
    a0 = np.arange(10).reshape(5, 2)
    a_pad = fft._pad(a0, (9, 11))
    assert_array_equal(a_pad, fftconv.pad_to(a0, (9, 11)))
    assert_array_equal(fft._centered(a_pad, (5, 2)), a0)
    assert fft._get_fft_shape((5, 58, 48), (5, 43, 43), 3, (1, 2)) == [108, 96]
    assert fft._get_fft_shape((5, 128, 128), (1, 41, 41), 3, (1, 2)) == [180, 180]
    assert fft._get_fft_shape((6, 40, 59), (6, 31, 31), 3, (1, 2)) == [75, 96]
    for n in (1, 7, 97, 172, 1000):
        assert fft.next_fast_len(n) == fftconv.next_fast_len(n)


def test_fft_psf_matching_like_reference():
    # reference tests/test_fft.py:91-123 + golden outputs of the reference
    g = golden("fft_psf")
    psf1 = fft.Fourier(scarlet.GaussianPSF(1, boxsize=41).get_model())
    psf2 = fft.Fourier(scarlet.GaussianPSF(2, boxsize=41).get_model())
    assert_allclose(psf1.image, g["psf1"], atol=1e-15)
    k12 = fft.match_psf(psf2, psf1)
    assert_almost_equal(fft.convolve(psf1, k12).image, psf2.image)
    assert_allclose(k12.image, g["k12"], atol=1e-12)
    k21 = fft.match_psf(psf1, psf2)
    assert_almost_equal(fft.convolve(psf2, k21).image, psf1.image)
    psf123 = fft.Fourier(scarlet.GaussianPSF((1, 2, 3), boxsize=41).get_model())
    assert_allclose(psf123.image, g["psf123"], atol=1e-15)
    km = fft.match_psf(psf123, psf1)
    assert_almost_equal(psf123.image, fft.convolve(km, psf1).image)
    for img in fft.convolve(fft.match_psf(psf1, psf123), psf123).image:
        assert_almost_equal(img, psf1.image[0])
    assert_allclose(fft.shift(g["shift_in"], (0.3, -1.7), return_Fourier=False), g["shift_out"],
                    atol=1e-6)
    assert_allclose(fft.convolve(g["cube"], g["kern"], axes=(1, 2), return_Fourier=False),
                    g["conv"], atol=1e-6)


def test_diff_kernel_matches_reference_golden(hsc):
    psf = scarlet.ImagePSF(hsc["psfs"].copy())
    model_psf = scarlet.GaussianPSF(sigma=(0.8,) * 5)
    assert_allclose(model_psf.get_model(), hsc["model_psf"], atol=1e-15)
    frame = scarlet.Frame(hsc["images"].shape, psf=model_psf, channels=list("grizy"))
    obs = scarlet.Observation(hsc["images"], psf=psf, weights=hsc["weights"],
                              channels=list("grizy")).match(frame)
    assert isinstance(obs.renderer, scarlet.ConvolutionRenderer)
    assert_allclose(obs.renderer.diff_kernel.image, hsc["diff_kernel"], atol=1e-7)
    assert_allclose(obs.log_norm, hsc["log_norm"], rtol=1e-7)
    assert_allclose(np.array(np.mean(obs.noise_rms, axis=(1, 2))), hsc["noise_rms_mean"], rtol=1e-6)
    same = scarlet.Observation(hsc["images"], psf=model_psf, channels=list("grizy")).match(frame)
    assert isinstance(same.renderer, scarlet.NullRenderer)


def test_monotonic_tables_match_reference():
    g = golden("operator_tables")
    for key in g.files:
        if key.startswith("w_"):
            _, mode, tag = key.split("_")
            h, w = map(int, tag.split("x"))
            mine = operator.getRadialMonotonicWeights((h, w), mode, (h // 2, w // 2))
            assert_allclose(mine, g[key], rtol=0, atol=1e-15)
        elif key.startswith("didx_"):
            h, w = map(int, key[5:].split("x"))
            assert_array_equal(operator.sort_by_radius((h, w), (h // 2, w // 2)), g[key])
    # default centre and the helper tables of the reference API
    assert_allclose(operator.getRadialMonotonicWeights((7, 9), "angle"),
                    proxops.radial_monotonic_weights((7, 9), "angle"), atol=1e-15)
    table, missing = operator.diagonalizeArray(np.arange(12.0).reshape(3, 4))
    ref_table, ref_missing = proxops._diagonalize(np.arange(12.0).reshape(3, 4))
    assert_array_equal(missing, ref_missing)
    assert_array_equal(table[~missing], ref_table[~ref_missing])
    offsets, sl, sl_inv = operator.getOffsets(4)
    assert offsets == [-5, -4, -3, -1, 1, 3, 4, 5]


@pytest.mark.parametrize("shape", [(3, 3), (5, 5), (21, 21), (31, 31), (41, 41), (47, 47), (49, 49),
                                   (51, 51), (61, 61), (63, 63), (65, 65), (81, 81), (95, 95),
                                   (31, 41), (22, 30), (40, 40), (41, 40), (7, 47), (46, 11)])
def test_ring_schedule_of_the_sweep_is_the_sequential_loop(shape):
    """The ring plan (library builder, csrc/sweep_plan.cpp) run by a numpy model of the
    device loop gives the bits of the reference's sequential loop
    (operators_pybind11.cc:14-36 via the oracle) for the three weightings, the default and a
    shifted peak (fit_center plans)."""
    import ring_model

    rng = np.random.default_rng(shape[0] * 100 + shape[1])
    h, w = shape
    for kind in ("angle", "flat", "nearest"):
        for centre in ((h // 2, w // 2), (min(h // 2 + 1, h - 1), max(w // 2 - 1, 0))):
            weights, offsets, didx = operator.monotonic_tables(shape, kind, centre)
            plan = ring_model.ring_plan(shape, weights, offsets, didx)
            rmax = max(centre[0], h - 1 - centre[0], centre[1], w - 1 - centre[1])
            assert (plan is not None) == (rmax <= 47)
            if plan is None:
                continue
            assert plan["planes"] == (1 if rmax <= 31 else 2)
            # rings 24 .. 31 share the one plane by starting late (csrc/common.h): 3 levels a
            # ring instead of 2 from ring 24 on
            late = max(0, rmax - 23) if rmax <= 31 else 0
            assert plan["n_steps"] == 3 * rmax - 1 + late
            assert plan["n_nat"] == (42 if late else plan["n_pad"])
            # the device stream stores the weights once per ring: only where all eight octants
            # hold the same pixels with bit-identical weights (centred odd squares)
            regular = h == w and h % 2 == 1 and centre == (h // 2, w // 2)
            steps = plan["n_pad"] + 6
            assert plan["stream_bytes"] == (steps * plan["planes"] * 256 if regular else 0)
            for g in (0.0, 0.25):
                img = rng.random(h * w).astype(np.float32)
                img[rng.integers(0, h * w, 5)] = 0  # exact zeros among the inputs
                ref = proxops.sweep(img.copy(), weights, offsets, didx, g)
                got = ring_model.run(plan, img, g)
                assert_array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_ring_schedule_is_refused_for_other_tables():
    """Tables without the radial structure (a weighted neighbour farther out, a sweep order
    that is not by radius) have no ring plan; the kernels keep the level plan for them."""
    import ring_model

    shape = (21, 21)
    weights, offsets, didx = operator.monotonic_tables(shape, "angle", (10, 10))
    assert ring_model.ring_plan(shape, weights, offsets, didx) is not None
    outward = weights.copy()
    p = 3 * 21 + 4
    outward[:, p] = 0
    outward[0, p] = 1.0  # towards the corner: away from the peak
    assert ring_model.ring_plan(shape, outward, offsets, didx) is None
    assert ring_model.ring_plan(shape, weights, offsets, didx[::-1].copy()) is None
    assert ring_model.ring_plan(shape, weights, offsets[::-1].copy(), didx) is None


def test_elementwise_constraints_like_reference():
    # reference tests/test_constraint.py:9-71, 137-171
    rng = np.random.default_rng(0)
    X = rng.random(100) - 0.5
    assert np.all(scarlet.PositivityConstraint()(X, 0) >= 0)
    assert np.all(scarlet.PositivityConstraint(zero=0.1)(X, 0) >= 0.1)
    Y = rng.random(100)
    assert_almost_equal(scarlet.NormalizationConstraint("sum")(Y.copy(), 0), Y / Y.sum())
    assert_almost_equal(scarlet.NormalizationConstraint("max")(Y.copy(), 0), Y / Y.max())
    step, thresh = 0.5, 0.25
    for typ, t in (("relative", thresh * step), ("absolute", thresh)):
        out = scarlet.L0Constraint(thresh=thresh, type=typ)(X.copy(), step)
        mask = np.abs(X) < t
        assert np.all(out[mask] == 0)
        assert_array_equal(out[~mask], X[~mask])
        out = scarlet.L1Constraint(thresh=thresh, type=typ)(X.copy(), step)
        assert np.all(out[mask] == 0)
        assert_array_equal(np.abs(out[~mask]), np.abs(np.abs(X[~mask]) - t))
    Z = np.arange(25, dtype=float).reshape(5, 5)
    assert_almost_equal(scarlet.SymmetryConstraint()(Z.copy(), 0), np.full((5, 5), 12.0))
    assert_almost_equal(scarlet.SymmetryConstraint(strength=0.5)(Z.copy(), 0), Z * 0.5 + 6)
    assert scarlet.CenterOnConstraint()(np.zeros((5, 5)), 0)[2, 2] > 0
    np.random.seed(0)
    noise = np.random.rand(21, 21) * 2
    signal = np.zeros(noise.shape)
    psf = scarlet.GaussianPSF(sigma=1, boxsize=21).get_model()
    signal[7:14, 7:14] = psf[0, 7:14, 7:14]
    Xn = signal + noise
    out = scarlet.ThresholdConstraint()(Xn.copy(), 0)
    mask = Xn < 0.05704869232578929
    assert np.all(out[mask] == 0)
    assert_array_equal(out[~mask], Xn[~mask])


def test_device_chain_translation():
    frame = scarlet.Frame((2, 30, 30), channels=[0, 1], psf=scarlet.GaussianPSF(0.8))
    morph = scarlet.ExtendedSourceMorphology(frame, (15, 15), np.ones((21, 21)),
                                             bbox=scarlet.Box((21, 21), origin=(5, 5)))
    f = device_flags(morph.parameters[0].constraint)
    assert f["flags"] == _lib.PROX_EXTENDED_SOURCE
    assert f["neighbor_weight"] == "angle" and f["min_gradient"] == 0
    sym = scarlet.ExtendedSourceMorphology(frame, (15, 15), np.ones((21, 21)),
                                           bbox=scarlet.Box((21, 21), origin=(5, 5)),
                                           monotonic="flat", symmetric=True, min_grad=0.1)
    f = device_flags(sym.parameters[0].constraint)
    assert f["flags"] == _lib.PROX_EXTENDED_SOURCE | _lib.PROX_SYMMETRY
    assert f["neighbor_weight"] == "flat" and f["min_gradient"] == 0.1
    assert device_flags(scarlet.PositivityConstraint())["flags"] == _lib.PROX_POSITIVE
    assert device_flags(None)["flags"] == 0
    with pytest.raises(NotImplementedError):
        device_flags(scarlet.ConstraintChain(scarlet.PositivityConstraint(),
                                             scarlet.MonotonicityConstraint()))
    with pytest.raises(NotImplementedError):
        device_flags(scarlet.Constraint(lambda x, s: x))


def test_component_placement_like_reference():
    # reference tests/test_component.py
    frame = scarlet.Frame((10, 20, 30), channels=np.arange(10))
    shape, on, origin = (5, 4, 6), (1, 2, 3), (2, 3, 4)
    sed = np.zeros(5)
    sed[on[0]] = 1
    morph = np.zeros(shape[1:])
    morph[on[1:]] = 1
    box = scarlet.Box(shape, origin=origin)
    comp = scarlet.FactorizedComponent(
        frame, scarlet.TabulatedSpectrum(frame, sed, bbox=box[0]),
        scarlet.ImageMorphology(frame, morph, bbox=box[1:]))
    model = comp.get_model(frame=frame)
    loc = tuple(np.array(on) + np.array(origin))
    mask = np.zeros(model.shape, dtype=bool)
    mask[loc] = True
    assert_array_equal(model[~mask], 0)
    assert model[loc] == 1
    assert [p.name for p in comp.parameters] == ["spectrum", "image", "shift"]
    cube = np.zeros(shape)
    cube[on] = 1
    cc = scarlet.CubeComponent(frame, scarlet.Parameter(cube, name="cube"), bbox=box)
    assert cc.get_model(frame=frame)[loc] == 1
    combined = scarlet.CombinedComponent([comp, comp])
    assert combined.get_model(frame=frame)[loc] == 2
    blend = scarlet.Blend([comp, combined],
                          scarlet.Observation(np.zeros(frame.shape, np.float32), channels=np.arange(10)))
    assert blend.get_model()[loc] == 3 and blend.get_model().dtype == np.float32
    assert len(blend.parameters) == 9


def test_parameter_pickles_with_state():
    p = scarlet.Parameter(np.arange(4.0), name="x", step=0.5, m=np.ones(4), fixed=True)
    q = pickle.loads(pickle.dumps(p))
    assert_array_equal(q, p)
    assert q.name == "x" and q.step == 0.5 and q.fixed and np.all(q.m == 1)
    assert scarlet.relative_step(np.array([1.0, 3.0]), 0, factor=0.5, minimum=0.2) == 1.0
    bad = scarlet.Parameter(np.array([np.nan]), name="bad")
    assert not bad.is_finite


def test_shard_ranges_cover_everything():
    from scarlet_amd import dist

    for n, world in ((1024, 8), (10, 3), (7, 8), (0, 2)):
        ranges = [dist.shard_range(n, r, world) for r in range(world)]
        assert ranges[0][0] == 0 and ranges[-1][1] == n
        for (a0, a1), (b0, b1) in zip(ranges, ranges[1:]):
            assert a1 == b0 and a0 <= a1
        sizes = [b - a for a, b in ranges]
        assert max(sizes) - min(sizes) <= 1


def test_measure_functions():
    """scarlet.measure names (reference measure.py): known answers on a small cube"""
    from scarlet_amd import measure

    cube = np.zeros((2, 5, 7))
    cube[0, 1, 2] = 3.0
    cube[1, 3, 4] = 1.0
    assert measure.max_pixel(cube) == (0, 1, 2)
    assert_allclose(measure.flux(cube), [3.0, 1.0])
    assert_allclose(measure.centroid(cube), [0.25, 1.5, 2.5])
    M = measure.moments(cube, N=2, centroid=(0, 0))
    assert set(M) == {(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)}
    assert_allclose(M[0, 0], [3.0, 1.0])
    # first index = power of the second-axis offset (the reference's convention)
    assert_allclose(M[1, 0], [3.0 * 2, 1.0 * 4])
    assert_allclose(M[0, 1], [3.0 * 1, 1.0 * 3])


def test_lite_host_utilities():
    """scarlet.lite helpers that never touch the GPU"""
    from scarlet_amd import lite

    psf = lite.integrated_circular_gaussian(sigma=0.8)
    assert psf.shape == (15, 15) and abs(psf.sum() - 1) < 1e-12 and psf.argmax() == 7 * 15 + 7
    assert lite.get_circle_mask(5).sum() == 13 and lite.get_circle_mask(4).sum() == 12
    box = lite.bounds_to_bbox((3, 7, 2, 4))
    assert box.shape == (5, 3) and box.origin == (3, 2)
    img = lite.insert_image(scarlet.Box((4, 6)), scarlet.Box((2, 2), origin=(1, 3)), np.ones((2, 2)))
    assert img.sum() == 4 and img[1, 3] == 1 and img[0, 0] == 0
    p = lite.FistaParameter(np.ones((5, 5)), step=0.5)
    p.grow((9, 9), 2)
    assert p.x.shape == p.z.shape == (9, 9) and p.x.sum() == 25
    p.shrink(3)
    assert p.x.shape == (3, 3)
    a = lite.AdaproxParameter(np.ones((5, 5), np.float32), step=1e-2)
    assert a.vhat.min() == -np.inf and a.step(a.x, 0) == 1e-2 and a.b1[7] == 0.9
    with pytest.raises(NotImplementedError):
        a.update(0, None)
    psfs = np.stack([lite.integrated_circular_gaussian(sigma=s) for s in (0.8, 0.8001, 0.8)])
    assert lite.get_min_psf(psfs, thresh=0.01).shape[1] < 15


def test_multiresolution_frames_and_renderer_setup():
    """Frame.from_observations with WCSs and the ResolutionRenderer set-up (host side of
    BASELINE config 5) reproduce the reference's model frames, PSFs and FFT shapes for
    every pair of tests/test_multiresolution.py (golden); the WCS conversion round-trips"""
    g = golden("multiresolution")

    def wcs(k):
        w = scarlet.LinearWCS(g["crpix_%d" % k], g["crval_%d" % k], g["pc_%d" % k], g["cdelt_%d" % k])
        w.array_shape = g["crpix_%d" % k] * 2
        return w

    w = wcs(0)
    pix = np.array([[0.0, 0.0], [10.5, 99.25], [130.0, 7.0]])
    assert_allclose(w.world_to_pixel_values(w.pixel_to_world_values(pix)), pix, atol=1e-9)
    # a subset here (the set-up of the big frames takes seconds each); the GPU test
    # walks through all twenty
    for tag in ("1_3_union", "1_4_intersection", "2_4_intersection", "3_4_union", "2_3_union"):
        i, j, coverage = str(tag).split("_")
        i, j = int(i), int(j)
        obs_hr = scarlet.Observation(g["image_%d" % i][None], wcs=wcs(i),
                                     psf=scarlet.ImagePSF(g["psf_%d" % i]), channels=["lr"])
        obs_lr = scarlet.Observation(g["image_%d" % j][None], wcs=wcs(j),
                                     psf=scarlet.ImagePSF(g["psf_%d" % j]), channels=["hr"])
        frame = scarlet.Frame.from_observations([obs_lr, obs_hr], obs_id=1, coverage=coverage)
        assert tuple(frame.shape) == tuple(g["frame_shape_%s" % tag]), tag
        assert_allclose(frame.wcs.wcs.crpix, g["frame_crpix_%s" % tag])
        assert_allclose(frame.psf.get_model(), g["model_psf_%s" % tag], atol=1e-12)
        assert type(obs_lr.renderer) is scarlet.ResolutionRenderer
        assert type(obs_hr.renderer).__name__ == str(g["hr_renderer_%s" % tag])
        assert list(obs_lr.renderer._fft_shape) == list(g["fft_shape_%s" % tag])
        n_lr = g["image_%d" % j].shape[0]
        assert obs_lr.renderer._resconv_op.shape == (1, n_lr, int(np.prod(g["fft_shape_%s" % tag])))


def test_the_reference_shift_operator_is_circulant_and_the_spectral_form_is_the_same_map():
    """What the library's spectral evaluation of the resampling operator rests on
    (resample.hip): the x-shift the reference applies to the padded model
    (renderer.py:414-476) is a circulant matrix, and with transforms along x Parseval's
    identity gives A . (model . Pt) and its transpose exactly.  Host operators of a fixture
    pair, float64 NumPy; no GPU."""
    g = golden("multiresolution")

    def wcs(k):
        w = scarlet.LinearWCS(g["crpix_%d" % k], g["crval_%d" % k], g["pc_%d" % k], g["cdelt_%d" % k])
        w.array_shape = g["crpix_%d" % k] * 2
        return w

    i, j = 1, 4
    obs_hr = scarlet.Observation(g["image_%d" % i][None], wcs=wcs(i),
                                 psf=scarlet.ImagePSF(g["psf_%d" % i]), channels=["lr"])
    obs_lr = scarlet.Observation(g["image_%d" % j][None], wcs=wcs(j),
                                 psf=scarlet.ImagePSF(g["psf_%d" % j]), channels=["hr"])
    scarlet.Frame.from_observations([obs_lr, obs_hr], obs_id=1, coverage="intersection")
    r = obs_lr.renderer
    Fy, Fx = r._fft_shape
    n_b = r.other_shifts.shape[1]
    P = r._shift_along(np.eye(Fx)[None], -r.other_shifts[1], axis=2)[0]  # [x', x, b]
    kern = P[0]  # s_b[x] = P[0, x, b]
    idx = (np.arange(Fx)[None, :] - np.arange(Fx)[:, None]) % Fx  # [x', x] -> (x - x') mod Fx
    assert np.abs(P - kern[idx]).max() < 1e-14 * np.abs(P).max()
    P32 = P.astype(np.float32)
    assert np.array_equal(P32, P32[0][idx])  # what smi_resampler_create checks
    A = r._resconv_op[0].reshape(-1, Fy, Fx).astype(np.float64)
    rng = np.random.default_rng(3)
    model = np.zeros((Fy, Fx))
    model[10:70, 25:90] = rng.random((60, 65))
    dense = np.einsum("ayx,yxb->ab", A, np.einsum("yp,pxb->yxb", model, P))
    Kx = Fx // 2 + 1
    w = np.full(Kx, 2.0)
    w[0] = 1.0
    if Fx % 2 == 0:
        w[-1] = 1.0
    E = np.fft.rfft(A, axis=2)
    beta = w * np.fft.rfft(kern, axis=0).T / Fx  # [b, k]
    G = np.einsum("ayk,yk->ak", np.conj(E), np.fft.rfft(model, axis=1))
    assert np.abs(np.real(G @ beta.T) - dense).max() < 1e-12 * np.abs(dense).max()
    # the transposed chain
    resid = rng.normal(size=dense.shape)
    g_dense = np.einsum("ab,ayx,pxb->yp", resid, A, P)
    Mbar = np.einsum("ak,ayk->yk", resid @ np.conj(beta), E)
    W = np.exp(-2j * np.pi * np.outer(np.arange(Fx), np.arange(Kx)) / Fx)
    g_spec = Mbar.real @ W.real.T + Mbar.imag @ W.imag.T
    assert np.abs(g_spec - g_dense).max() < 1e-12 * np.abs(g_dense).max()


def test_tan_wcs_matches_astropy():
    """Gnomonic WCS against astropy's conversions (golden: the two cut-outs of the
    multi-resolution tutorial, whose reference pixels lie 1800 to 30000 pixels outside the
    images, sampled inside and around the images).  The flat-sky LinearWCS is NOT good
    enough there, which is why TanWCS exists."""
    from conftest import golden
    from scarlet_amd.wcs import LinearWCS, TanWCS

    g = golden("multires_tutorial")
    for tag in ("hsc", "hst"):
        args = (g["crpix_" + tag], g["crval_" + tag], g["pc_" + tag], g["cdelt_" + tag])
        w = TanWCS(*args)
        sky = w.pixel_to_world_values(g["sample_pix_" + tag])
        assert np.abs(sky - g["sample_sky_" + tag]).max() < 1e-12  # degrees
        back = w.world_to_pixel_values(g["sample_sky_" + tag])
        assert np.abs(back - g["sample_back_" + tag]).max() < 1e-7  # pixels
        assert np.abs(back - g["sample_pix_" + tag]).max() < 1e-7
        flat = LinearWCS(*args).world_to_pixel_values(g["sample_sky_" + tag])
        assert np.abs(flat - g["sample_back_" + tag]).max() > 0.1
        # header round trip and the facade's interface
        header = {"CRPIX1": args[0][0], "CRPIX2": args[0][1], "CRVAL1": args[1][0],
                  "CRVAL2": args[1][1], "CD1_1": args[2][0][0], "CD1_2": args[2][0][1],
                  "CD2_1": args[2][1][0], "CD2_2": args[2][1][1], "NAXIS1": 50, "NAXIS2": 40}
        h = TanWCS.from_header(header)
        assert h.array_shape == (40, 50) and h.celestial is h
        assert np.abs(h.pixel_to_world_values(g["sample_pix_" + tag]) - sky).max() < 1e-12
        moved = h.deepcopy()
        moved.wcs.crpix -= (3, 5)
        assert np.abs(moved.world_to_pixel_values(sky) - (back - (3, 5))).max() < 1e-7


def test_host_stepped_parameter_matches_a_float64_restatement():
    """hoststep.HostParameter (the host's AMSGrad + prox for user constraints): the
    float32 arithmetic agrees with a plain float64 restatement of
    lite/parameters.py:274-305; ``wave_sum`` is the 64-lane tree"""
    from scarlet_amd import Parameter, PositivityConstraint, hoststep, relative_step
    from functools import partial

    rng = np.random.default_rng(3)
    x = rng.uniform(1, 5, 1681).astype(np.float32)
    assert abs(float(hoststep.wave_sum(x)) - float(x.astype(np.float64).sum())) < 1e-3
    tree = x[:64].copy()
    while tree.size > 1:
        tree = tree[0::2] + tree[1::2]
    assert hoststep.wave_sum(x[:64]) == tree[0]

    sed = Parameter(rng.uniform(1, 5, 5).astype(np.float32), name="spectrum",
                    step=partial(relative_step, factor=1e-2, minimum=0.05),
                    constraint=PositivityConstraint(1e-20))
    hp = hoststep.HostParameter(sed, "sed", (0.0, 1e-2, 0.05))
    x64 = np.asarray(sed, dtype=np.float64).copy()
    m = v = vh = np.zeros(5)
    for it in range(4):
        g = rng.normal(size=5)
        alpha = max(0.05, 1e-2 * x64.mean())
        m = 0.1 * g + 0.9 * m
        v = 0.001 * g * g + 0.999 * v
        vh = v.copy() if it == 0 else np.maximum(vh, v)
        psi = np.sqrt(np.maximum(vh, 1e-8))
        x64 = np.maximum(x64 - alpha * m / psi / (10 if it == 0 else 1), 1e-20)
        hp.update(it, g.astype(np.float32), 1e-3, 10, 0.9, 0.999, 1e-8)
        assert np.allclose(np.asarray(sed), x64, rtol=1e-5)
    hp.store()
    # 1 - float32(0.999) differs from 0.001 by 5e-5 relative (the device's arithmetic)
    assert sed.m.dtype == np.float64 and np.allclose(sed.v, v, rtol=2e-4)

    # a prior adds what it returns for the current value to the gradient (blend.py:120-131)
    from scarlet_amd import Prior

    class Pull(Prior):
        def __call__(self, x):
            return 0.3 * (x - 2.0)

        def grad(self, x):
            return 0.3 * np.ones_like(x)

    prior = Parameter(rng.uniform(1, 5, 5).astype(np.float32), name="spectrum", step=0.05,
                      prior=Pull(), constraint=PositivityConstraint(1e-20))
    hp = hoststep.HostParameter(prior, "sed", (0.05, 0.0, 0.0))
    x64 = np.asarray(prior, dtype=np.float64).copy()
    m = v = vh = np.zeros(5)
    for it in range(3):
        g = rng.normal(size=5)
        gp = g + 0.3 * (x64 - 2.0)
        m = 0.1 * gp + 0.9 * m
        v = 0.001 * gp * gp + 0.999 * v
        vh = v.copy() if it == 0 else np.maximum(vh, v)
        x64 = np.maximum(x64 - 0.05 * m / np.sqrt(np.maximum(vh, 1e-8)) / (10 if it == 0 else 1), 1e-20)
        hp.update(it, g.astype(np.float32), 1e-3, 10, 0.9, 0.999, 1e-8)
        assert np.allclose(np.asarray(prior), x64, rtol=1e-5)


def test_adaprox_schemes_of_the_host_step():
    """``Blend.fit(scheme=...)`` other than AMSGrad steps on the host (hoststep.phi_psi).
    Adam against a float64 restatement of Kingma & Ba; AdamX with a constant b1 and PAdam
    with p = 1/2 are AMSGrad; RAdam takes plain momentum steps while rho_t <= 4."""
    from scarlet_amd import Parameter, hoststep

    rng = np.random.default_rng(5)
    x0 = rng.uniform(1, 2, 7).astype(np.float32)
    grads = rng.normal(size=(8, 7)).astype(np.float32)

    def run(scheme, **kw):
        x = Parameter(x0.copy(), name="x", step=0.05)
        hp = hoststep.HostParameter(x, "sed", (0.05, 0.0, 0.0), scheme, **kw)
        trace = []
        for it, g in enumerate(grads):
            hp.update(it, g, 1e-3, 10, 0.9, 0.999, 1e-8)
            trace.append(np.array(x))
        return np.array(trace)

    with pytest.raises(ValueError):
        hoststep.HostParameter(Parameter(x0.copy(), name="x", step=0.05), "sed", (0.05, 0, 0), "sgd")
    ams = run("amsgrad")
    assert np.array_equal(run("adamx"), ams)
    assert np.allclose(run("padam", p=0.5), ams, rtol=1e-6)
    assert not np.allclose(run("padam"), ams, rtol=1e-3)

    x, m, v = x0.astype(np.float64), np.zeros(7), np.zeros(7)
    adam = run("adam")
    for it, g in enumerate(grads.astype(np.float64)):
        m = 0.1 * g + 0.9 * m
        v = 0.001 * g * g + 0.999 * v
        t = it + 1
        step = 0.05 * (m / (1 - 0.9 ** t)) / (np.sqrt(v / (1 - 0.999 ** t)) + 1e-8)
        x = x - step / (10 if it == 0 else 1)
        assert np.allclose(adam[it], x, rtol=2e-4)

    # rho_t ~ t for small t at b2 = 0.999: psi = 1, phi = bias-corrected momentum up to t = 4
    radam = run("radam")
    x, m = x0.astype(np.float64), np.zeros(7)
    for it, g in enumerate(grads[:3].astype(np.float64)):
        m = 0.1 * g + 0.9 * m
        x = x - 0.05 * m / (1 - 0.9 ** (it + 1)) / (10 if it == 0 else 1)
        assert np.allclose(radam[it], x, rtol=1e-5)
    assert np.all(np.isfinite(run("nadam"))) and not np.allclose(run("nadam"), adam)


def test_bench_prices_the_bytes_of_survey_8d():
    """bench.algorithmic_bytes is SURVEY.md 8d's accounting: the nominated figures of config
    2 / 3 (F = 180^2: B0 1 194 880 B, B0 + B_fft 6 669 760 B) and config 1 (2 778 560 B), and
    the per-kernel split adds up to the whole plus the gather of the gradient image."""
    import importlib.util
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cfg3 = bench.algorithmic_bytes(5, 128, 128, [41 * 41] * 10, 180, 180, 1)
    assert cfg3["null"] == 1194880 and cfg3["whole"] == 6669760
    boxes = [41 * 41] * 5 + [61 * 61] * 2 + [31 * 31] + [21 * 21] * 2
    cfg1 = bench.algorithmic_bytes(5, 58, 48, boxes, 108, 96, 5)
    assert cfg1["null"] == 679040 and cfg1["whole"] == 2778560
    for by, bx in ((cfg3, [41 * 41] * 10), (cfg1, boxes)):
        gather = 4 * 5 * sum(bx)
        assert by["conv"] + by["update"] == by["whole"] + gather
    run = bench.algorithmic_bytes(5, 128, 128, [41 * 41] * 10, 160, 160, 1)
    # roofline.frac_survey_model prices the 8d model, roofline.frac the bytes a kernel must
    # move at its fusion boundary: data + weights + parameters in, gradient image out
    assert run["conv"] == 5051760 and run["update"] == 808280
    assert run["conv_boundary"] == 1050480 and run["update_boundary"] == run["update"]
    assert run["conv_boundary"] + run["update_boundary"] - 4 * 5 * 10 * 41 * 41 \
        - 4 * 5 * 128 * 128 == run["null"]  # = B0 + the gradient image written once and gathered under the boxes


def test_resized_component_description_is_what_specs_would_make(hsc):
    """fit_blends derives the device description of a resized component from the old one
    (blend._resized_spec: new image, new origin, halved step); it must be field for field what
    Blend._specs makes from scratch after ImageMorphology.update shrank or grew the box."""
    import scarlet_amd as scarlet
    from scarlet_amd.blend import _flatten, _resized_spec
    from scarlet_amd.model import UpdateException
    from test_gpu_facade import build_blend

    blend, _ = build_blend(hsc, resizing=True)
    comps = _flatten(blend.sources)
    before = blend._specs(comps)
    assert not blend._host
    changed = 0
    for k, comp in enumerate(comps):
        morphology = comp.children[1]
        image = morphology.parameters[0]
        h, w = image.shape
        if k % 2:   # make it shrink: nothing but the centre above zero
            image[...] = 0
            image[h // 2, w // 2] = 1
        else:       # make it grow: a strong pull on one edge
            image.m = np.zeros(image.shape)
            image.v = np.ones(image.shape)
            image.m[:, 0] = -1e3
            image[...] = 1
        try:
            morphology.update()
        except UpdateException:
            changed += 1
            fresh = blend._specs([comp])[0]
            derived = _resized_spec(before[k], comp)
            assert derived.morph.shape != before[k].morph.shape
            for name, want in vars(fresh).items():
                got = getattr(derived, name)
                if isinstance(want, np.ndarray):
                    assert_array_equal(got, want, err_msg=name)
                else:
                    assert got == want, name
    assert changed >= 6


    """morphology._empty_margin against the loop of the reference's shrink_box
    (morphology.py:50-67): peel while all four outermost rows / columns hold nothing above
    the threshold."""
    from scarlet_amd.morphology import _empty_margin

    def loop(image, thresh):
        dist = 0  # (the reference's loop; it ends before the middle when a pixel is occupied)
        while (np.all(image[dist, :] <= thresh) and np.all(image[-dist - 1, :] <= thresh)
               and np.all(image[:, dist] <= thresh) and np.all(image[:, -dist - 1] <= thresh)):
            dist += 1
        return dist

    rng = np.random.default_rng(3)
    for trial in range(300):
        n = int(rng.integers(3, 40)) | 1
        image = np.zeros((n, n), dtype=np.float32)
        k = int(rng.integers(0, n // 2 + 1))
        if k < n - k:
            image[k:n - k, k:n - k] = rng.random((n - 2 * k, n - 2 * k)) - 0.2
        got = _empty_margin(image, 0)
        if (image > 0).any():
            assert got == loop(image, 0), (n, k)
        else:
            assert got == (n + 1) // 2


def test_edge_pull_of_the_resize_hook_is_the_masked_array_formula():
    """morphology._edge_pull (four edge slices, plain arrays) gives the bits of the
    reference's expression on the whole image as a masked array (morphology.py:166-176),
    including edges where the second moment is zero in places or everywhere."""
    import warnings

    import numpy.ma as ma
    from scarlet_amd.morphology import _edge_pull

    rng = np.random.default_rng(11)
    for shape in ((21, 21), (41, 41), (31, 41), (5, 3), (61, 61)):
        for trial in range(6):
            image = (rng.random(shape) - 0.3).astype(np.float32)
            m = rng.standard_normal(shape).astype(np.float32).astype(np.float64)
            v = (rng.random(shape) ** 4).astype(np.float32).astype(np.float64)
            v[rng.random(shape) < 0.2] = 0
            if trial == 1:
                v[:, 0] = 0            # one edge without a single unmasked pixel
            if trial == 2:
                v[...] = 0
            step = 1e-2 / 2 ** trial
            gu = -m / np.sqrt(np.sqrt(ma.masked_equal(v, 0))) * step
            pull = gu * (image > 0)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")  # "converting a masked element to nan"
                want = np.array((pull[:, 0].mean(), pull[:, -1].mean(),
                                 pull[0, :].mean(), pull[-1, :].mean()))
            got = _edge_pull(image, m, v, step)
            assert_array_equal(np.isnan(got), np.isnan(want))
            ok = ~np.isnan(want)
            assert_array_equal(got[ok].view(np.uint64), want[ok].view(np.uint64))


def test_bench_names_the_dominant_kernel_from_the_library_path():
    """The kernel the roofline object prices follows the convolution path the library reports
    and the larger of the two phase times -- not a threshold on an empty phase's event
    overhead (a 128-blend shard: conv 0.10 ms, conv_adj 5 us of events, update 0.13 ms)."""
    import importlib.util
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    by = bench.algorithmic_bytes(5, 128, 128, [41 * 41] * 10, 160, 160, 1)
    shard = {"render": 0.0052, "conv": 0.0994, "residual": 0.0047, "conv_adj": 0.0053,
             "update": 0.1329, "total": 0.2474}
    name, nbytes, ms = bench.dominant_kernel(shard, "fused", by)
    assert name.startswith("update kernels") and nbytes == by["update"] and ms == 0.1329
    full = dict(shard, conv=0.6511, update=0.3353, total=1.0017)
    assert bench.dominant_kernel(full, "fused", by) == ("fused_conv_kernel", by["conv"], 0.6511)
    assert bench.dominant_kernel(full, "rocfft", by)[0] == "whole iteration (rocFFT pipeline)"
    assert bench.dominant_kernel(full, "rocfft", by)[1:] == (by["whole"], 1.0017)
    assert bench.dominant_kernel(full, "none", by) == ("update_kernel_reg", by["update"], 0.3353)


def test_importing_the_package_asks_for_eight_hardware_queues():
    """A fresh process that says nothing about GPU_MAX_HW_QUEUES gets 8 exported by the
    import (before the HIP runtime can have started); a value of the caller's stays, and
    SCARLET_AMD_HW_QUEUES=keep leaves the environment alone."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, sys; sys.path.insert(0, %r); import scarlet_amd; "
            "print(os.environ.get('GPU_MAX_HW_QUEUES'), scarlet_amd.configure())" % root)

    def run(**env):
        base = {k: v for k, v in os.environ.items()
                if k not in ("GPU_MAX_HW_QUEUES", "SCARLET_AMD_HW_QUEUES")}
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True,
                             env=dict(base, **env), timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        return out.stdout.split()[-2:]

    assert run() == ["8", "8"]
    assert run(GPU_MAX_HW_QUEUES="2") == ["2", "2"]
    assert run(SCARLET_AMD_HW_QUEUES="keep") == ["None", "4"]


def test_configure_sets_the_hardware_queues_only_on_request(monkeypatch):
    """``configure(hw_queues=8)`` sets GPU_MAX_HW_QUEUES while the HIP runtime has not started,
    warns and changes nothing once it has, and the library is told the number in effect
    (``load()`` itself leaves the environment alone)."""
    import os
    import warnings

    import scarlet_amd
    from scarlet_amd import _lib

    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    monkeypatch.setattr(_lib, "_hw_queues", None)
    monkeypatch.setattr(_lib, "_warned", False)
    # (a process whose environment was left alone: SCARLET_AMD_HW_QUEUES=keep)
    monkeypatch.setattr(_lib, "_env_at_import", None)
    lib = _lib.load()
    assert "GPU_MAX_HW_QUEUES" not in os.environ
    assert scarlet_amd.configure() == 4
    # the runtime is up: nothing changes, one warning
    monkeypatch.setattr(_lib, "_hip_started", lambda: True)
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        assert scarlet_amd.configure(hw_queues=8) == 4
        assert scarlet_amd.configure(hw_queues=8) == 4
    assert len(seen) == 1 and "already started" in str(seen[0].message)
    assert "GPU_MAX_HW_QUEUES" not in os.environ
    # ... unless the variable was exported before (the runtime has read it)
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "8")
    assert scarlet_amd.configure(hw_queues=8) == 8
    assert lib.smi_set_hw_queues(8) == 8
    # before the runtime starts: set on request
    monkeypatch.delenv("GPU_MAX_HW_QUEUES")
    monkeypatch.setattr(_lib, "_hw_queues", None)
    monkeypatch.setattr(_lib, "_hip_started", lambda: False)
    assert scarlet_amd.configure(hw_queues=8) == 8 and os.environ["GPU_MAX_HW_QUEUES"] == "8"
    assert lib.smi_set_hw_queues(4) == 8  # (back to the default for the other tests)


def test_set_spectra_to_match_reproduces_the_reference_on_the_host():
    """``initialization.set_spectra_to_match`` on the quickstart scene, from the reference's
    own initial morphologies (golden): the per-band least squares -- float64 renders of the
    unit-spectrum components, float64 normal equations, like initialization.py:493-588 --
    gives the reference's spectra to float32 rounding.  (Host code: it needs no GPU.)"""
    import scarlet_amd as scarlet
    from scarlet_amd.initialization import set_spectra_to_match
    from conftest import golden

    hsc = golden("hsc_cosmos_35")
    filters = list("grizy")
    frame = scarlet.Frame(hsc["images"].shape, psf=scarlet.GaussianPSF(sigma=(0.8,) * 5),
                          channels=filters)
    obs = scarlet.Observation(hsc["images"], psf=scarlet.ImagePSF(hsc["psfs"]),
                              weights=hsc["weights"], channels=filters).match(frame)
    comps = []
    for k in range(int(hsc["n_comp"])):
        h, w = hsc["morph_%d" % k].shape
        oy, ox = hsc["origin_%d" % k]
        box = scarlet.Box((5, h, w), origin=(0, int(oy), int(ox)))
        comps.append(scarlet.FactorizedComponent(
            frame, scarlet.TabulatedSpectrum(frame, np.ones(5, dtype=np.float32), bbox=box[0]),
            scarlet.ImageMorphology(frame, hsc["morph_%d" % k].astype(np.float64), bbox=box[1:])))
    set_spectra_to_match(comps, obs)
    for k, c in enumerate(comps):
        sed, ref = np.asarray(c.children[0].parameters[0]), hsc["sed_%d" % k]
        assert np.abs(sed - ref).max() < 2e-6 * np.abs(ref).max(), k


def test_lazy_std_survives_views_and_old_pickles():
    """``Parameter.std`` after a fit is a marker that turns into the estimate on first access
    (round-4 advice): views, slices and comparisons of the Parameter carry the marker on
    without making the masked array, and a pickle from before ``std`` was a property loads."""
    import pickle

    from scarlet_amd.parameter import Parameter, STD_FROM_V

    p = Parameter(np.arange(6, dtype=np.float64).reshape(2, 3), name="image", step=1e-2)
    p.v = np.full((2, 3), 4.0)
    p.std = STD_FROM_V
    q = p[0]
    _ = p > 0
    assert p.__dict__["_std"] is STD_FROM_V and q.__dict__["_std"] is STD_FROM_V
    assert np.allclose(p.std, 0.5) and not isinstance(p.__dict__["_std"], str)
    # legacy pickle: the attribute dictionary holds 'std'
    base = p.__reduce__()
    legacy = dict(base[2][-1])
    legacy["std"] = legacy.pop("_std")
    r = Parameter(np.zeros((2, 3)), name="x")
    r.__setstate__(base[2][:-1] + (legacy,))
    assert np.allclose(r.std, 0.5)
    assert np.allclose(pickle.loads(pickle.dumps(p)).std, 0.5)
