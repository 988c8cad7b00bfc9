"""world_size-2 run of the sharding + result-gather path on CPU (gloo).  The fit
itself needs a GPU; here every rank produces a deterministic stand-in for the
per-blend records of its shard so that the collective part is exercised."""

import os
import subprocess
import sys
import textwrap

from conftest import ROOT

SCRIPT = textwrap.dedent(
    """
    import sys
    import numpy as np
    sys.path.insert(0, %r)
    from scarlet_amd import dist
    rank, local_rank, world = dist.init_process_group(backend="gloo")
    assert world == 2
    n_total = 11                                   # ragged on purpose: 6 + 5
    lo, hi = dist.shard_range(n_total, rank, world)
    ids = np.arange(lo, hi)
    n_iter, logL = dist.gather_results(10 + ids, -1000.0 - ids)
    assert n_iter.tolist() == list(range(10, 10 + n_total)), n_iter
    assert np.allclose(logL, -1000.0 - np.arange(n_total))
    assert dist.max_over_ranks(1.0 + rank) == 2.0
    dist.barrier()
    sys.stdout.write("rank" + str(rank) + "-ok" + chr(10))
    sys.stdout.flush()
    """
) % ROOT


def test_gather_over_two_ranks(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(SCRIPT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29611", str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "rank0-ok" in out.stdout and "rank1-ok" in out.stdout, out.stdout
