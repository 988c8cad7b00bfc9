"""Multi-GPU driver: independent blends shard across ranks (one process per
GPU); the only communication is the gather of per-blend results at the end
(RCCL through ``torch.distributed`` backend "nccl" on GPUs, "gloo" in the CPU
tests).  There is no collective inside the iteration loop -- the reference has
no cross-blend state at all."""

import os

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


def gather_results(n_iter, logL, device=None):
    """All-gather the per-blend result records ``{n_iter:int32, logL:float64}``
    of every rank; shards may have different lengths.  Returns (n_iter, logL)
    for the whole job in global blend order on every rank."""
    import torch
    import torch.distributed as dist

    n_iter = np.asarray(n_iter, dtype=np.int32)
    logL = np.asarray(logL, dtype=np.float64)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return n_iter, logL
    world = dist.get_world_size()
    if device is None:
        device = "cuda" if dist.get_backend() == "nccl" else "cpu"
    counts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([n_iter.size], dtype=torch.int64, device=device))
    counts = [int(c.item()) for c in counts]
    width = max(counts) if counts else 0
    rec = torch.zeros((width, 2), dtype=torch.float64, device=device)
    rec[: n_iter.size, 0] = torch.from_numpy(n_iter.astype(np.float64)).to(device)
    rec[: n_iter.size, 1] = torch.from_numpy(logL).to(device)
    out = [torch.zeros_like(rec) for _ in range(world)]
    dist.all_gather(out, rec)
    out = [o.cpu().numpy()[:c] for o, c in zip(out, counts)]
    allrec = np.concatenate(out, axis=0) if out else np.zeros((0, 2))
    return allrec[:, 0].astype(np.int32), allrec[:, 1]


def max_over_ranks(value, device=None):
    """MAX-reduce a Python float over all ranks (timing)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    if device is None:
        device = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
