"""Worker of tests/test_sharding.py: one rank of a world_size-2 gloo group on CPU. Checks that the shards of the
synthetic batch tile the global batch exactly and that the timing reduction is a max over ranks."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from acvm_amd import shard, synth  # noqa: E402


def main():
    rank, _, world = shard.env_rank()
    dist = shard.init_group(rank, world)
    per_gpu = 96
    first, last = shard.shard_range(rank, world, per_gpu)
    mine = synth.witness_batch(per_gpu, seed=0xAC1D0002, first_instance=first)
    whole = synth.witness_batch(per_gpu * world, seed=0xAC1D0002)
    n_in = 16
    assert mine == whole[first * n_in * 32:last * n_in * 32], "shard is not a slice of the global batch"
    # every rank contributes a digest of its shard; all ranks see the same set
    d = torch.tensor(list(hashlib.sha256(mine).digest()), dtype=torch.uint8)
    got = [torch.zeros_like(d) for _ in range(world)]
    dist.all_gather(got, d)
    for r in range(world):
        f, l = shard.shard_range(r, world, per_gpu)
        assert bytes(got[r].tolist()) == hashlib.sha256(whole[f * n_in * 32:l * n_in * 32]).digest()
    shard.barrier(dist)
    elapsed = 1.0 + rank  # rank 1 is the slow one
    assert shard.max_over_ranks(elapsed, dist) == float(world)
    assert shard.sum_over_ranks(per_gpu, dist) == per_gpu * world
    assert shard.split_total(10, 3) == [(0, 4), (4, 7), (7, 10)]
    dist.destroy_process_group()
    print(f"rank {rank} ok")


if __name__ == "__main__":
    main()
