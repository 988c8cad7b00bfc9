"""Does the driver's 18-ms window run at the clocks a longer run reaches?  Ten windows of
`--steps` iterations back to back (state restored in between, on the device), after an idle
pause like the one bench.py's set-up leaves: blend-iterations/s of each.

    python tools/clock_ramp.py [--blends 1024] [--steps 20] [--idle 2.0]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blends", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--idle", type=float, default=2.0)
    args = ap.parse_args()
    import scarlet_amd
    scarlet_amd.configure(hw_queues=8)
    import torch
    import bench
    from scarlet_amd import BlendBatch

    data, weights, comps, kernel, _ = bench.build_cfg3(0, args.blends, 0, None)
    batch = BlendBatch(data, weights, comps, kernel=kernel, max_iter=args.steps + 1, device=0)
    batch.save_state()
    for round_ in range(2):
        time.sleep(args.idle)  # the GPU idles (bench.py: scene construction, oracle check)
        batch.restore_state()
        batch.step(0, 5, e_rel=1e-3, check_convergence=False)  # the driver's warm-up
        torch.cuda.synchronize()
        rates = []
        for _ in range(10):
            batch.restore_state()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            batch.step(0, args.steps, e_rel=1e-3, check_convergence=False)
            torch.cuda.synchronize()
            rates.append(args.blends * args.steps / (time.perf_counter() - t0))
        print("after %.1f s idle + 5 warm-up iterations: " % args.idle
              + " ".join("%.0fk" % (r / 1e3) for r in rates), flush=True)
    batch.close()


if __name__ == "__main__":
    main()
