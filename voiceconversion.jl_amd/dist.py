"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).

The hot path shards embarrassingly: frames for fvconvert, pairs for DTW, utterances for trajectory
conversion -- no data-path collective.  Only the E-step has an exchange step: ONE all-reduce(sum) of the packed
sufficient statistics [S0 | S1 | S2 | loglik] (M(1+2Dj)+1 doubles, 165 KB at Dj=80, M=128: latency-bound)."""
import os


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init_process_group(backend=None):
    """Initialise torch.distributed from the torchrun environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).
    backend defaults to "nccl" (RCCL) when a HIP device is visible, else "gloo"."""
    import torch
    import torch.distributed as dist

    rank, world, local = env_rank_world()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local)
        from . import _lib

        _lib.set_device(local)
    if world > 1 and not dist.is_initialized():
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(n, rank, world):
    """Contiguous, balanced [lo, hi) block of n items for `rank` (frame sharding of a (D,n) matrix: a contiguous
    column block is a contiguous byte range of the Julia memory image)."""
    base, rem = divmod(int(n), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_by_cost(costs, world):
    """Longest-processing-time partition of items with the given costs (e.g. S*T per DTW pair, T per utterance)
    into `world` index lists with balanced total cost.  Deterministic."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads = [0] * world
    parts = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        parts[r].append(i)
        loads[r] += costs[i]
    return [sorted(p) for p in parts]


def allreduce_sum_(t, group=None):
    """In-place sum over ranks; a no-op without an initialised process group.  A one-rank group still goes through the
    backend (RCCL on a GPU): that is how the collective path is exercised on a 1-GPU box."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def pack_stats(S0, S1, S2, loglik):
    """(S0 (M,), S1 (Dj,M), S2 (Dj,M), loglik) -> the packed layout of vcmi_estep_diag_dev, as a float64 numpy vector."""
    import numpy as np

    return np.concatenate([np.asarray(S0, dtype=np.float64).ravel(), np.asarray(S1, dtype=np.float64).ravel(order="F"),
                           np.asarray(S2, dtype=np.float64).ravel(order="F"), np.array([loglik], dtype=np.float64)])
