"""Tiles inside one device for batches whose witness table exceeds HBM: one batch handle of `tile`
instances is reused for consecutive slices of the global batch; per tile only the per-instance results and the witnesses
the caller asks for (normally the circuit's return values, Circuit.witness_set("return_values")) are kept.

Footprint of a handle = 32 B x witnesses x tile: a 10^6-opcode circuit runs with tile = 4096 (131 GB of the 288 GB); the
instances of different tiles are independent, exactly like the instances of different GPUs (acvm_amd/shard.py)."""
import numpy as np

from . import Batch


def solve_tiled(circuit, initial_ids, values_be: bytes, n_instances: int, tile: int, keep_witnesses, solver=None, digests=None):
    """Solves n_instances instances (values_be = [n_instances][len(initial_ids)][32] big-endian) in tiles of `tile`.
    Returns (results, values): results = list of per-instance Result, values = uint8 array [n_instances][len(keep)][32] with
    zeros for instances that did not solve (their return witnesses may be unassigned). `digests`: optional uint8 array
    [n_instances][32] that receives the per-instance digest of the full witness map (Batch.digest) -- what SURVEY 8d keeps of a
    config-5 tile besides the return witnesses."""
    ids = list(initial_ids)
    keep = list(keep_witnesses)
    row = len(ids) * 32
    if len(values_be) != n_instances * row:
        raise ValueError("values_be has the wrong size")
    tile = max(1, min(tile, n_instances)) if n_instances else 1
    batch = Batch(circuit, tile, ids, solver)
    results = []
    out = np.zeros((n_instances, len(keep), 32), dtype=np.uint8)
    try:
        for first in range(0, n_instances, tile):
            n = min(tile, n_instances - first)
            chunk = values_be[first * row:(first + n) * row]
            if n < tile:  # the last tile is padded with copies of its first instance; the padding is dropped below
                chunk = chunk + chunk[:row] * (tile - n)
            batch.set_initial_witness(chunk)
            batch.solve()
            res = batch.results()[:n]
            results.extend(res)
            if digests is not None:
                digests[first:first + n] = batch.digest(0, n)
            if keep:
                asg, vals = None, None
                solved = [i for i in range(n) if res[i].status == 0]
                if len(solved) == n:
                    out[first:first + n] = batch.extract(keep, 0, n)
                else:  # extract_indices refuses unassigned witnesses: take the solved instances one run at a time
                    i = 0
                    while i < n:
                        if res[i].status != 0:
                            i += 1
                            continue
                        k = i
                        while k < n and res[k].status == 0:
                            k += 1
                        out[first + i:first + k] = batch.extract(keep, i, k - i)
                        i = k
    finally:
        batch.free()
    return results, out
