"""Experiment: does splitting the 1024-blend batch into S sub-batches on S streams hide the
tail of the update kernel behind the other sub-batches' convolution kernel?  Blends are
independent, so sub-batch A's iteration i+1 only waits for A's update i.

    python tools/two_stream_overlap.py [n_streams ...]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import build_scenes  # noqa: E402
from scarlet_amd.batch import BlendBatch, ComponentSpec  # noqa: E402

NB, WARM, STEPS = 1024, 10, 50
kern, scenes = build_scenes(NB, 1234)


def run(n_streams, chunk):
    per = NB // n_streams
    batches, streams = [], []
    for s in range(n_streams):
        sub = scenes[s * per:(s + 1) * per]
        comps = [[ComponentSpec(sc["seds"][k], sc["morphs"][k], sc["origins"][k],
                                sed_min_step=sc["noise_rms"]) for k in range(len(sc["morphs"]))]
                 for sc in sub]
        b = BlendBatch(np.stack([sc["data"] for sc in sub]), np.stack([sc["weights"] for sc in sub]),
                       comps, kernel=kern[2], max_iter=WARM + STEPS + 1)
        st = torch.cuda.Stream()
        b.set_stream(st.cuda_stream)
        batches.append(b)
        streams.append(st)
    for b in batches:
        b.step(0, WARM, e_rel=1e-3, prox_max_iter=10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(WARM, WARM + STEPS, chunk):
        for b in batches:
            b.step(it, min(chunk, WARM + STEPS - it), e_rel=1e-3, prox_max_iter=10)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    logL = np.concatenate([[-l[-1] for l in b.loss_history()] for b in batches])
    for b in batches:
        b.close()
    return NB * STEPS / dt, 1e3 * dt / STEPS, logL.mean()


for arg in (sys.argv[1:] or ["1", "2", "4"]):
    n = int(arg)
    for chunk in (1, 5):
        v, ms, logL = run(n, chunk)
        print("streams %d, %d iteration(s) per launch group: %.0f blend-it/s, %.3f ms per iteration, mean logL %.3f"
              % (n, chunk, v, ms, logL))


def run_lib(n_sub, chunk):
    """the same through the library's own sub-ranges (smi_batch_set_sub_ranges)"""
    comps = [[ComponentSpec(sc["seds"][k], sc["morphs"][k], sc["origins"][k],
                            sed_min_step=sc["noise_rms"]) for k in range(len(sc["morphs"]))]
             for sc in scenes]
    b = BlendBatch(np.stack([sc["data"] for sc in scenes]), np.stack([sc["weights"] for sc in scenes]),
                   comps, kernel=kern[2], max_iter=WARM + STEPS + 1)
    st = torch.cuda.Stream()
    b.set_stream(st.cuda_stream)
    b.set_sub_ranges(n_sub)
    b.step(0, WARM, e_rel=1e-3, prox_max_iter=10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(WARM, WARM + STEPS, chunk):
        b.step(it, min(chunk, WARM + STEPS - it), e_rel=1e-3, prox_max_iter=10)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    b.close()
    return NB * STEPS / dt, 1e3 * dt / STEPS


for n in (1, 2, 3, 4, 6, 8):
    for chunk in (STEPS, 10):
        v, ms = run_lib(n, chunk)
        print("library sub-ranges %d, %d iterations per call: %.0f blend-it/s, %.3f ms per iteration"
              % (n, chunk, v, ms))
