"""The multi-resolution operators with random sizes (non-square FFT shapes, n_a != n_b,
sizes off the GEMM tiles) and random operators -- dense ones (two products per band) and circulant shift operators
(the spectral path) in turn --, GPU against NumPy in float64:
smi_resampler_render, and the low-resolution term's loss and gradient through
smi_batch_attach_lowres on a NullRenderer batch.  Development aid.

    python tools/fuzz_resampler.py [n_cases] [seed]      (FUZZ_BIG=1: frames of 100 .. 260 pixels)
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from scarlet_amd import _lib  # noqa: E402
from scarlet_amd.batch import BlendBatch, ComponentSpec  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 13)
lib = _lib.load()
bad, worst = [], dict(render=0.0, loss=0.0, grad=0.0)
for n in range(n_cases):
    C = int(rng.integers(1, 4))
    H, W = int(rng.integers(8, 70)), int(rng.integers(8, 70))
    Fy, Fx = H + int(rng.integers(0, 30)), W + int(rng.integers(0, 30))
    n_a, n_b = int(rng.integers(1, 40)), int(rng.integers(1, 40))
    if os.environ.get("FUZZ_BIG"):  # several 64-lane chunks of k, more than 64 columns b
        C = 1
        H, W = int(rng.integers(100, 220)), int(rng.integers(130, 260))
        Fy, Fx = H + int(rng.integers(0, 40)), W + int(rng.integers(0, 40))
        n_a, n_b = int(rng.integers(30, 110)), int(rng.integers(50, 90))
    A = rng.normal(0, 1, (C, n_a, Fy * Fx)).astype(np.float32)
    P = rng.normal(0, 1, (Fx, n_b, Fx)).astype(np.float32)  # P[x, b, x']
    if n % 2:  # a circulant shift operator (what the reference builds): the spectral path
        kern = rng.normal(0, 1, (n_b, Fx)).astype(np.float32)
        idx = (np.arange(Fx)[:, None] - np.arange(Fx)[None, :]) % Fx  # [x, x']
        P = np.ascontiguousarray(kern[:, idx].transpose(1, 0, 2))
    Pt = np.ascontiguousarray(P.transpose(2, 0, 1).reshape(Fx, Fx * n_b))
    handle = ctypes.c_void_p()
    _lib.check(lib.smi_resampler_create(_lib.ptr(A, ctypes.c_float), _lib.ptr(Pt, ctypes.c_float),
                                        C, n_a, n_b, Fy, Fx, ctypes.byref(handle)))
    path = ctypes.c_int32(-1)
    _lib.check(lib.smi_resampler_get_path(handle, ctypes.byref(path)))
    assert path.value == n % 2, "path %d for case %d" % (path.value, n)
    desc = "C=%d frame %dx%d F=%dx%d n_a=%d n_b=%d path=%d" % (C, H, W, Fy, Fx, n_a, n_b, path.value)

    def render64(padded):
        shifted = np.einsum("cyz,xbz->cyxb", padded.astype(np.float64), P.astype(np.float64))
        return np.einsum("cak,ckb->cab", A.astype(np.float64),
                         shifted.reshape(C, Fy * Fx, n_b))

    try:
        padded = rng.normal(0, 1, (C, Fy, Fx)).astype(np.float32)
        out = np.empty((C, n_a, n_b), np.float32)
        _lib.check(lib.smi_resampler_render(handle, _lib.ptr(padded, ctypes.c_float),
                                            _lib.ptr(out, ctypes.c_float)))
        ref = render64(padded)
        dev = dict(render=np.abs(out - ref).max() / np.abs(ref).max())
        # the term in a batch: one component covering the frame, no convolution, the
        # high-resolution observation carries no weight
        Cm = C + 1  # one model channel the low-resolution observation does not see
        channels = sorted(rng.choice(Cm, C, replace=False).tolist())
        sed = rng.uniform(0.5, 2, Cm).astype(np.float32)
        # (a component box the update kernels hold in LDS: the whole frame up to 70 pixels)
        bh, bw = min(H, 70), min(W, 70)
        oy, ox = int(rng.integers(0, H - bh + 1)), int(rng.integers(0, W - bw + 1))
        morph = rng.random((bh, bw)).astype(np.float32)
        data = rng.normal(0, 1, (C, n_a, n_b)).astype(np.float32)
        weights = rng.uniform(0.5, 2, (C, n_a, n_b)).astype(np.float32)
        weights[rng.random(weights.shape) < 0.1] = 0
        batch = BlendBatch(np.zeros((1, Cm, H, W), np.float32), np.zeros((1, Cm, H, W), np.float32),
                           [[ComponentSpec(sed, morph, (oy, ox), prox_flags=0)]], kernel=None, max_iter=2)
        batch.attach_lowres(handle, channels, data, weights, 1.25)
        _, _, logL = batch.forward()
        g_sed, g_morph = batch.gradient()
        batch.close()
        model = np.zeros((Cm, H, W))
        model[:, oy:oy + bh, ox:ox + bw] = sed[:, None, None].astype(np.float64) * morph[None].astype(np.float64)
        y0, x0 = (Fy - H + 1) // 2, (Fx - W + 1) // 2
        pad = np.zeros((C, Fy, Fx))
        pad[:, y0:y0 + H, x0:x0 + W] = model[channels]
        resid = weights * (render64(pad) - data)
        loss = 1.25 + 0.5 * np.sum(resid * (render64(pad) - data))
        back = np.einsum("cak,cab->ckb", A.astype(np.float64), resid).reshape(C, Fy, Fx, n_b)
        gpad = np.einsum("cyxb,xbz->cyz", back, P.astype(np.float64))
        G = np.zeros((Cm, H, W))
        G[channels] = gpad[:, y0:y0 + H, x0:x0 + W]
        G = G[:, oy:oy + bh, ox:ox + bw]
        ref_sed = np.einsum("cyx,yx->c", G, morph)
        ref_morph = np.einsum("c,cyx->yx", sed, G)
        dev["loss"] = abs(-logL[0] - loss) / abs(loss)
        scale_s = np.einsum("cyx,yx->c", np.abs(G), morph).max()
        scale_m = np.einsum("c,cyx->yx", sed, np.abs(G)).max()
        dev["grad"] = max(np.abs(g_sed[0] - ref_sed).max() / scale_s,
                          np.abs(g_morph[0] - ref_morph).max() / scale_m)
    finally:
        lib.smi_resampler_destroy(handle)
    for key, val in dev.items():
        worst[key] = max(worst[key], float(val))
    over = {k: float(v) for k, v in dev.items() if v > 2e-5}
    if over:
        bad.append((n, desc, over))
    if os.environ.get("FUZZ_VERBOSE"):
        print(n, desc, {k: "%.1e" % v for k, v in dev.items()})
print("resampler cases: %d; worst deviations: %s" % (n_cases, {k: "%.2e" % v for k, v in worst.items()}))
for entry in bad:
    print("OVER", entry)
