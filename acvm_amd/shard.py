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


# ---- digest of digests: proof that the ranks solved disjoint shards of ONE global batch (independent of the number of ranks
# and of the tile size). The per-instance map digests (acvm_batch_digest) of DIGEST_CHUNK consecutive instances of the global
# batch are hashed into one chunk digest; the digest of digests is Blake2s-256 over the chunk digests in global order. Shards
# are multiples of the chunk, so every rank produces whole chunks and rank 0 only concatenates them.
DIGEST_CHUNK = 1 << 12


def chunk_digests(instance_digests, chunk=DIGEST_CHUNK):
    """instance_digests: uint8 array [n][32] of consecutive instances starting at a multiple of `chunk` -> list of 32-byte chunk
    digests (the last one may cover fewer instances)."""
    import hashlib
    import numpy as np
    d = np.ascontiguousarray(instance_digests, dtype=np.uint8)
    return [hashlib.blake2s(d[i:i + chunk].tobytes()).digest() for i in range(0, d.shape[0], chunk)]


def digest_of_digests(chunks) -> str:
    import hashlib
    return hashlib.blake2s(b"".join(chunks)).hexdigest()


def gather_bytes(payload: bytes, dist):
    """all ranks' byte strings (equal length on every rank), in rank order; [payload] when single-process"""
    if dist is None:
        return [payload]
    import torch
    t = torch.tensor(list(payload), dtype=torch.uint8)
    got = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(got, t)
    return [bytes(g.tolist()) for g in got]


def gather_floats(value: float, dist):
    if dist is None:
        return [value]
    import torch
    t = torch.tensor([value], dtype=torch.float64)
    got = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(got, t)
    return [float(g.item()) for g in got]


def cpu_budget():
    """(CPUs this process may actually use, the cgroup's CPU quota or None, os.cpu_count()): os.cpu_count() is the HOST's (256 on the GPU box of
    this pool) while the container's cgroup grants a quota (cpu.max "1600000 100000" = 16 CPUs there) -- threads beyond the quota are only
    throttled, which is what made 64 threads look like 13 x one thread in round 4's line"""
    host = os.cpu_count() or 1
    try:
        host = min(host, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    usable = host if quota is None else max(1, min(host, int(quota + 0.5)))
    return usable, quota, os.cpu_count() or 1
