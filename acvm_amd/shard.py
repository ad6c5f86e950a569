"""Instance sharding for multi-GPU runs: the path shards by witness instance with no exchange step.
Rank r of n owns the contiguous instances [r * per_gpu, (r + 1) * per_gpu) of the global batch; the circuit plan is
replicated. torch.distributed is used only for the timing barrier and the max-over-ranks reduction (gloo: the data path
has no collective, so no RCCL communicator is created)."""
import os


def env_rank():
    """(rank, local_rank, world_size) from the torch.distributed.run environment."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_range(rank: int, world: int, per_gpu: int):
    """[first, last) global instance indices of a rank (weak scaling: per-GPU batch fixed)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return rank * per_gpu, (rank + 1) * per_gpu


def split_total(total: int, world: int):
    """Contiguous near-equal split of a fixed global batch (strong scaling): list of (first, last)."""
    base, rem = divmod(total, world)
    out, first = [], 0
    for r in range(world):
        n = base + (1 if r < rem else 0)
        out.append((first, first + n))
        first += n
    return out


def init_group(rank: int, world: int):
    """gloo process group for the barrier / max reduction; None when single-process."""
    if world <= 1:
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if not dist.is_initialized():
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    return dist


def barrier(dist):
    if dist is not None:
        dist.barrier()


def max_over_ranks(value: float, dist) -> float:
    if dist is None:
        return value
    import torch
    t = torch.tensor([value], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, dist) -> float:
    if dist is None:
        return value
    import torch
    t = torch.tensor([value], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
