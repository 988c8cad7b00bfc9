"""Multi-GPU driver: independent blends shard across ranks (one process per GPU).

Partition (BASELINE.json configs[2]): the job's blends
``0 .. n-1`` are cut into contiguous chunks, rank ``g`` of ``G`` owns
``blends[g n/G : (g+1) n/G]`` (``shard_range``).  There is no collective inside the
iteration loop -- the reference has no cross-blend state at all; its unit is the
per-blend loop of ``scarlet/testing/api.py:216-224``.  The only communication is the
all-gather of one packed result record per blend at the end

    {n_iter: i32, converged: i32, logL: f64, loss_hist[max_iter]: f64}

(what ``Blend.fit`` returns and leaves in ``blend.loss``, blend.py:189-194, 273) and,
on request, of the final parameters.  RCCL through ``torch.distributed`` backend
"nccl" on GPUs, "gloo" in the CPU tests and when ranks share one GPU.
"""

import os
import socket

import numpy as np


def env_rank():
    """(rank, local_rank, world_size) from the torchrun environment."""
    return (
        int(os.environ.get("RANK", "0")),
        int(os.environ.get("LOCAL_RANK", "0")),
        int(os.environ.get("WORLD_SIZE", "1")),
    )


def shard_range(n_total, rank, world_size):
    """Contiguous block of blend indices owned by ``rank``:
    ``[rank n/G, (rank+1) n/G)`` with the remainder spread over the first ranks."""
    base, extra = divmod(int(n_total), int(world_size))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def free_port():
    """A TCP port nobody listens on right now (rendezvous of a self-launched job)."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_command(n_ranks, script, argv, port=None):
    """The ``torch.distributed.run`` command line that starts ``script argv`` as
    ``n_ranks`` ranks on this node (one rank per GPU)."""
    import sys

    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
            "--nproc-per-node", str(int(n_ranks)), "--master-addr", "127.0.0.1",
            "--master-port", str(port or free_port()), script] + list(argv)


def init_process_group(backend=None, device_index=None):
    """Initialise ``torch.distributed`` from the environment if WORLD_SIZE > 1."""
    import torch
    import torch.distributed as dist

    rank, local_rank, world = env_rank()
    if device_index is None:
        device_index = local_rank
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(device_index)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def _active(min_world=2):
    import torch.distributed as dist

    return dist.is_available() and dist.is_initialized() and dist.get_world_size() >= min_world


def _device(device=None):
    """Where the tensors of a collective live: the rank's GPU with RCCL (backend "nccl"),
    host memory with gloo."""
    import torch
    import torch.distributed as dist

    if device is not None:
        return device
    if dist.get_backend() == "nccl":
        return "cuda:%d" % torch.cuda.current_device()
    return "cpu"


def comm_info(device=None):
    """What the LIVE process group looks like, for a bench line to carry: backend
    ("nccl" = RCCL on ROCm), ``dist.get_world_size()`` and the device of every rank -- one
    all-gather of a short string per rank, so the record shows that the collective library
    really saw N ranks.  A single process reports a world of one without a group."""
    import torch

    name = None
    if torch.cuda.is_available():
        i = torch.cuda.current_device()
        name = "cuda:%d %s" % (i, torch.cuda.get_device_name(i))
    if not _active(1):
        return {"backend": None, "world_size": 1, "devices": [name]}
    import torch.distributed as dist

    mine = "%s pid %d" % (name, os.getpid())
    devices = [bytes(b).decode() for b in all_gather_bytes(mine.encode(), device=device,
                                                           single_rank_too=True)]
    return {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
            "devices": devices}


# ------------------------------------------------------------------ collectives
def all_gather_bytes(buf, device=None, single_rank_too=False):
    """All-gather one byte string per rank (lengths may differ).  Returns the list of
    ``np.uint8`` arrays in rank order, on every rank.  Two collectives: the lengths,
    then the payloads padded to the longest.  ``single_rank_too`` runs the collectives in
    a process group of one rank as well (a test's way through RCCL on a one-GPU box)."""
    import torch
    import torch.distributed as dist

    buf = np.frombuffer(bytes(buf), dtype=np.uint8) if not isinstance(buf, np.ndarray) else \
        np.ascontiguousarray(buf).view(np.uint8).reshape(-1)
    if not _active(1 if single_rank_too else 2):
        return [buf.copy()]
    world = dist.get_world_size()
    device = _device(device)
    counts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([buf.size], dtype=torch.int64, device=device))
    counts = [int(c.item()) for c in counts]
    width = max(max(counts), 1)
    mine = torch.zeros(width, dtype=torch.uint8, device=device)
    if buf.size:
        mine[: buf.size] = torch.from_numpy(buf.copy()).to(device)
    out = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(out, mine)
    return [o.cpu().numpy()[:c].copy() for o, c in zip(out, counts)]


def record_dtype(max_iter):
    """Packed per-blend result record of SURVEY.md 8e."""
    return np.dtype([("n_iter", "<i4"), ("converged", "<i4"), ("logL", "<f8"),
                     ("loss_hist", "<f8", (int(max_iter),))])


def pack_records(loss_histories, states, max_iter):
    """Records of this rank's blends from ``BlendBatch.loss_history()`` (list of 1-D
    arrays) and ``BlendBatch.states()`` (2 = the stopping rule fired)."""
    rec = np.zeros(len(loss_histories), dtype=record_dtype(max_iter))
    rec["loss_hist"] = np.nan
    for i, loss in enumerate(loss_histories):
        n = min(len(loss), int(max_iter))
        rec["n_iter"][i] = len(loss)
        rec["converged"][i] = int(states[i] == 2)
        rec["logL"][i] = -loss[-1] if len(loss) else np.nan
        rec["loss_hist"][i, :n] = loss[:n]
    return rec


def gather_records(records, device=None):
    """All-gather the ranks' record arrays (``pack_records``); the result holds the
    whole job in global blend order on every rank."""
    parts = all_gather_bytes(records.view(np.uint8).reshape(-1), device=device)
    return np.concatenate([p.view(records.dtype) for p in parts])


def gather_parameters(seds, morphs, device=None):
    """All-gather the final parameters of every component: ``seds`` (n_comp, C)
    float32 and the list of morphology images.  Returns (seds, list of images) of the
    whole job in global component order."""
    seds = np.ascontiguousarray(seds, dtype=np.float32)
    shapes = np.array([m.shape for m in morphs], dtype=np.int32).reshape(-1, 2)
    flat = np.concatenate([np.asarray(m, dtype=np.float32).reshape(-1) for m in morphs]) \
        if len(morphs) else np.zeros(0, np.float32)
    head = np.array([seds.shape[0], seds.shape[1] if seds.ndim == 2 else 0], dtype=np.int32)
    blob = np.concatenate([head.view(np.uint8), shapes.view(np.uint8).reshape(-1),
                           seds.view(np.uint8).reshape(-1), flat.view(np.uint8)])
    all_seds, all_morphs = [], []
    for part in all_gather_bytes(blob, device=device):
        n, C = (int(x) for x in part[:8].view(np.int32))
        if n == 0:
            continue
        o = 8
        shp = part[o:o + 8 * n].view(np.int32).reshape(n, 2)
        o += 8 * n
        all_seds.append(part[o:o + 4 * n * C].view(np.float32).reshape(n, C))
        o += 4 * n * C
        pix = part[o:].view(np.float32)
        at = 0
        for h, w in shp:
            all_morphs.append(pix[at:at + h * w].reshape(h, w).copy())
            at += h * w
    if not all_seds:
        return np.zeros((0, 0), np.float32), []
    return np.concatenate(all_seds), all_morphs


def gather_results(n_iter, logL, device=None):
    """(n_iter, logL) of every blend of the job in global order (the short form of
    ``gather_records`` for callers that do not keep loss histories)."""
    rec = np.zeros(len(n_iter), dtype=record_dtype(0))
    rec["n_iter"], rec["logL"] = np.asarray(n_iter, np.int32), np.asarray(logL, np.float64)
    out = gather_records(rec, device=device)
    return out["n_iter"].copy(), out["logL"].copy()


def gather_objects(obj):
    """All-gather one picklable Python object per rank (facade state of ``fit_blends``)."""
    import pickle

    return [pickle.loads(p.tobytes()) for p in all_gather_bytes(pickle.dumps(obj))]


def max_over_ranks(value, device=None):
    """MAX-reduce a Python float over all ranks (timing)."""
    import torch
    import torch.distributed as dist

    if not _active():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    import torch
    import torch.distributed as dist

    if _active():
        if dist.get_backend() == "nccl":  # name the device: the rank's communicator is bound to it
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


# ------------------------------------------------------------------ sharded fit
def fit_sharded(make_batch, n_total, max_iter=200, e_rel=1e-3, min_iter=1, prox_max_iter=10,
                sync_every=10, with_parameters=False, device=None):
    """Fit ``n_total`` independent blends over all ranks of the process group.

    ``make_batch(lo, hi)`` builds the ``BlendBatch`` of the global blends ``lo .. hi-1``
    on this rank's GPU.  Every rank fits its ``shard_range`` with the device loop
    (``BlendBatch.fit``: per-blend stopping rule on the device) and the records are
    all-gathered.  Returns the record array of the whole job (``record_dtype``), plus
    ``(seds, morphs)`` of all components with ``with_parameters``.  Results do not
    depend on the number of ranks."""
    rank, _, world = env_rank()
    if not _active():
        rank, world = 0, 1
    lo, hi = shard_range(n_total, rank, world)
    rec = np.zeros(0, dtype=record_dtype(max_iter))
    seds, morphs = np.zeros((0, 0), np.float32), []
    if hi > lo:
        batch = make_batch(lo, hi)
        try:
            batch.fit(max_iter=max_iter, e_rel=e_rel, min_iter=min_iter,
                      prox_max_iter=prox_max_iter, sync_every=sync_every)
            rec = pack_records(batch.loss_history(), batch.states(), max_iter)
            if with_parameters:
                seds, morphs = batch.parameters()
        finally:
            batch.close()
    out = gather_records(rec, device=device)
    if with_parameters:
        return out, gather_parameters(seds, morphs, device=device)
    return out
