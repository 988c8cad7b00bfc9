"""Development aid: the CPU oracle (NumPy/C port of the reference loop) on all host cores,
one blend per worker process -- the multi-core figure next to bench.py's single-thread
cpu_baseline.  Not part of bench.py (rocprofv3 does not get along with fork pools)."""
import multiprocessing as mp
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("OMP_NUM_THREADS", "1")
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")


def fit(seed, n_iter=60):
    from oracle import pgm
    from scarlet_amd import synthetic

    s = synthetic.make_blend(seed)
    comps = [pgm.Component(s["seds"][k].copy(), s["morphs"][k].copy(), s["origins"][k],
                           sed_min_step=s["noise_rms"]) for k in range(len(s["morphs"]))]
    sc = pgm.Scene(s["data"].shape, s["data"], s["weights"], s["diff_kernel"], comps)
    for it in range(n_iter):
        sc.step(it, 1e-3)
    return n_iter


if __name__ == "__main__":
    cores = os.cpu_count()
    n_blends = 2 * cores
    fit(1234, 2)  # warm-up (library loads)
    with mp.get_context("spawn").Pool(cores) as pool:
        pool.map(fit, range(1234, 1234 + cores))  # warm-up of the workers
        t0 = time.perf_counter()
        done = sum(pool.map(fit, range(2000, 2000 + n_blends)))
        dt = time.perf_counter() - t0
    print("cores %d: %d blend-iterations in %.1f s = %.0f blend-iterations/s" % (cores, done, dt, done / dt))
