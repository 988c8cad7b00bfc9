"""world_size-2 runs of the sharding + result-gather path on CPU (gloo).  The fit
itself needs a GPU (tests/test_gpu_dist.py runs it with two ranks on one GPU); here
every rank produces a deterministic stand-in for the per-blend records of its shard
so that the partition and the collectives are exercised: the packed SURVEY 8e record
{n_iter, converged, logL, loss_hist[max_iter]}, ragged parameter gathers, pickled
facade state and empty shards."""

import os
import subprocess
import sys
import textwrap

import numpy as np

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

    # the packed record: loss histories of different lengths, state 2 = converged
    max_iter = 7
    losses = [np.arange(1 + b %% max_iter, dtype=float) + 100 * b for b in ids]
    states = [2 if b %% 2 else 0 for b in ids]
    rec = dist.gather_records(dist.pack_records(losses, states, max_iter))
    assert rec.dtype == dist.record_dtype(max_iter) and len(rec) == n_total
    for b in range(n_total):
        n = 1 + b %% max_iter
        assert rec["n_iter"][b] == n and rec["converged"][b] == b %% 2
        assert rec["logL"][b] == -(n - 1 + 100 * b)
        assert np.array_equal(rec["loss_hist"][b, :n], np.arange(n) + 100.0 * b)
        assert np.all(np.isnan(rec["loss_hist"][b, n:]))

    # final parameters: 2 components per blend, box sizes depend on the blend
    seds = np.array([[b, k, 0.5] for b in ids for k in range(2)], dtype=np.float32)
    morphs = [np.full((3 + b %% 3, 4 + k), b + 0.25 * k, np.float32) for b in ids for k in range(2)]
    all_seds, all_morphs = dist.gather_parameters(seds, morphs)
    assert all_seds.shape == (2 * n_total, 3) and len(all_morphs) == 2 * n_total
    for b in range(n_total):
        for k in range(2):
            assert all_seds[2 * b + k].tolist() == [b, k, 0.5]
            m = all_morphs[2 * b + k]
            assert m.shape == (3 + b %% 3, 4 + k) and np.all(m == b + 0.25 * k)

    # more ranks than blends: rank 1 owns nothing
    lo1, hi1 = dist.shard_range(1, rank, world)
    rec1 = dist.gather_records(dist.pack_records([np.array([3.0, 2.0])] * (hi1 - lo1),
                                                 [2] * (hi1 - lo1), 4))
    assert len(rec1) == 1 and rec1["n_iter"][0] == 2 and rec1["logL"][0] == -2.0
    s1, m1 = dist.gather_parameters(np.ones((hi1 - lo1, 5), np.float32),
                                    [np.ones((2, 2), np.float32)] * (hi1 - lo1))
    assert s1.shape == (1, 5) and len(m1) == 1

    objs = dist.gather_objects({"rank": rank, "ids": ids.tolist()})
    assert [o["rank"] for o in objs] == [0, 1] and objs[1]["ids"] == list(range(6, 11))
    assert dist.max_over_ranks(1.0 + rank) == 2.0
    dist.barrier()

    # the branch a multi-GPU job takes: backend "nccl" (RCCL) -> every tensor of the
    # collectives on the rank's GPU.  No GPU here: the backend name is mocked and the
    # requested device "cuda:0" is recorded and mapped onto host memory.
    import torch
    import torch.distributed as td
    asked = []
    def on_host(device):
        if str(device).startswith("cuda"):
            asked.append(str(device))
            return "cpu"
        return device
    real = dict(zeros=torch.zeros, tensor=torch.tensor, to=torch.Tensor.to, backend=td.get_backend,
                current=torch.cuda.current_device, barrier=td.barrier)
    torch.zeros = lambda *a, device=None, **k: real["zeros"](*a, device=on_host(device), **k)
    torch.tensor = lambda *a, device=None, **k: real["tensor"](*a, device=on_host(device), **k)
    torch.Tensor.to = lambda self, device, *a, **k: real["to"](self, on_host(device), *a, **k)
    td.get_backend = lambda *a, **k: "nccl"
    torch.cuda.current_device = lambda: 0
    td.barrier = lambda device_ids=None: (asked.append("barrier%%s" %% device_ids), real["barrier"]())[1]
    try:
        assert dist._device() == "cuda:0"
        rec = dist.gather_records(dist.pack_records(losses, states, max_iter))
        assert len(rec) == n_total and rec["n_iter"][7] == 1 + 7 %% max_iter
        assert dist.max_over_ranks(5.0 - rank) == 5.0
        dist.barrier()
    finally:
        torch.zeros, torch.tensor, torch.Tensor.to = real["zeros"], real["tensor"], real["to"]
        td.get_backend, torch.cuda.current_device, td.barrier = real["backend"], real["current"], real["barrier"]
    # buffers for the lengths of all ranks, my length, my payload and its upload, the timing scalar
    assert asked.count("cuda:0") >= 4 + world and "barrier[0]" in asked, asked
    sys.stdout.write("rank" + str(rank) + "-ok" + chr(10))
    sys.stdout.flush()
    """
) % ROOT


def test_gather_over_two_ranks(tmp_path):
    from scarlet_amd import dist

    script = tmp_path / "worker.py"
    script.write_text(SCRIPT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = dist.launch_command(2, str(script), [])
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "rank0-ok" in out.stdout and "rank1-ok" in out.stdout, out.stdout


def test_shard_range_is_a_contiguous_partition():
    from scarlet_amd import dist

    for n in (0, 1, 7, 128, 1024, 1031):
        for world in (1, 2, 3, 4, 8):
            cuts = [dist.shard_range(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1
    # BASELINE configs[2]: 1024 blends over 8 GPUs = 128 contiguous blends each
    assert dist.shard_range(1024, 3, 8) == (384, 512)


def test_single_process_gathers_are_identities():
    from scarlet_amd import dist

    rec = dist.pack_records([np.array([5.0, 4.0, 3.5])], [2], 5)
    out = dist.gather_records(rec)
    assert out["n_iter"][0] == 3 and out["converged"][0] == 1 and out["logL"][0] == -3.5
    seds, morphs = dist.gather_parameters(np.ones((1, 5), np.float32), [np.zeros((3, 4))])
    assert seds.shape == (1, 5) and morphs[0].shape == (3, 4)
