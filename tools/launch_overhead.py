"""Is a small shard bound by the host's launch rate?  Enqueue time of BlendBatch.step (the call
returns when everything is queued) beside the time until the device is done.

    python tools/launch_overhead.py [--blends 128] [--steps 20]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blends", type=int, default=128)
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    import scarlet_amd
    scarlet_amd.configure(hw_queues=8)
    import torch
    import bench
    from scarlet_amd import BlendBatch

    data, weights, comps, kernel, _ = bench.build_cfg3(0, args.blends, 0, None)
    for ranges in (1, 2, 3, 4, 6, 8):
        batch = BlendBatch(data, weights, comps, kernel=kernel, max_iter=args.steps + 1, device=0)
        batch.set_sub_ranges(ranges)
        batch.save_state()
        batch.step(0, 5, e_rel=1e-3, check_convergence=False)
        torch.cuda.synchronize()
        best = None
        for _ in range(3):
            batch.restore_state()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            batch.step(0, args.steps, e_rel=1e-3, check_convergence=False)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            row = (t2 - t0, t1 - t0)
            if best is None or row[0] < best[0]:
                best = row
        print("blends %d ranges %d: total %.3f ms, enqueue %.3f ms (%.1f us per range-iteration), %.0f blend-it/s"
              % (args.blends, ranges, best[0] * 1e3, best[1] * 1e3,
                 best[1] * 1e6 / (args.steps * ranges), args.blends * args.steps / best[0]), flush=True)
        batch.close()


if __name__ == "__main__":
    main()
