"""Tiles inside one device for batches whose witness table exceeds HBM: one batch handle of `tile`
instances is reused for consecutive slices of the global batch; per tile only the per-instance results and the witnesses
the caller asks for (normally the circuit's return values, Circuit.witness_set("return_values")) are kept.

Footprint of a handle = 32 B x witnesses x tile: a 10^6-opcode circuit runs with tile = 4096 (131 GB of the 288 GB); the
instances of different tiles are independent, exactly like the instances of different GPUs (acvm_amd/shard.py)."""
import numpy as np

from . import Batch, DeviceBuffer


class ResidentShard:
    """One rank's shard of a global batch with its inputs resident in HBM (SURVEY 8d timing protocol: H2D outside the timed
    region): `n` instances, values_be = [n][len(ids)][32], solved as consecutive tiles of `tile` instances through ONE reused
    batch handle (acvm_batch_set_initial_witness_device on a slice of the resident buffer, then acvm_batch_solve). When tile
    does not divide n the last tile starts at n - tile and overlaps its predecessor (independent instances: solving one twice
    changes nothing)."""

    def __init__(self, circuit, initial_ids, values_be, n: int, tile: int, solver=None, resident=True):
        self.ids = list(initial_ids)
        self.row = len(self.ids) * 32
        self.n = n
        self.tile = max(1, min(tile, n))
        self.host = np.frombuffer(values_be, dtype=np.uint8)
        if self.host.size != n * self.row:
            raise ValueError("values_be has the wrong size")
        self.batch = Batch(circuit, self.tile, self.ids, solver)
        self.buf = DeviceBuffer(self.host) if resident else None
        self.starts = list(range(0, n - self.tile + 1, self.tile))
        if self.starts[-1] + self.tile < n:
            self.starts.append(n - self.tile)

    def tile_ptr(self, k: int) -> int:
        return self.buf.ptr + self.starts[k % len(self.starts)] * self.row

    def load_tile(self, k: int):
        """ACVM::new for tile k from the resident buffer (free when the previous solve_tile already brought it in)"""
        self.batch.set_initial_witness_device(self.tile_ptr(k))

    def solve_tile(self, k: int, pipelined=True) -> int:
        """ACVM::solve of tile k (loaded before); pipelined: the import of tile k + 1 (cyclically: the next pass starts over) rides behind the solve
        (acvm_batch_solve_then_import), so that the device does not idle across the tile boundary. Only for passes that read nothing of the
        initial witnesses between tiles."""
        return self.batch.solve(then_import=self.tile_ptr(k + 1) if pipelined else 0)

    def solve_pass(self, on_tile=None):
        """one pass over the shard; on_tile(k, first_new, first_in_tile) runs after tile k's solve, while its table is live:
        instances [first_new, start + tile) of the shard are the ones this tile solved for the first time"""
        done = 0
        for k, start in enumerate(self.starts):
            self.load_tile(k)
            self.batch.solve()
            if on_tile is not None:
                on_tile(k, done, start)
            done = start + self.tile

    def digests(self):
        """uint8 [n][32]: per-instance digests of the solved witness maps (one extra pass)"""
        out = np.zeros((self.n, 32), dtype=np.uint8)

        def grab(k, first_new, start):
            out[first_new:start + self.tile] = self.batch.digest(first_new - start, start + self.tile - first_new)
        self.solve_pass(grab)
        return out

    def free(self):
        self.batch.free()
        if self.buf is not None:
            self.buf.free()


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
    host = np.frombuffer(values_be, dtype=np.uint8)
    # H2D of tile k + 1 runs beside the solve of tile k: two device staging buffers, the upload on a helper thread (hipMemcpy
    # releases the GIL through ctypes)
    import threading
    stage = [DeviceBuffer(size=max(tile * row, 1)), DeviceBuffer(size=max(tile * row, 1))]

    from . import current_device, set_device
    dev = current_device()

    def upload(slot, first):
        set_device(dev)  # HIP's current device is per thread
        n = min(tile, n_instances - first)
        chunk = host[first * row:(first + n) * row]
        stage[slot].upload(chunk)
        if n < tile:  # the last tile is padded with copies of its first instance; the padding is dropped below
            pad = np.tile(chunk[:row], tile - n)
            stage[slot].upload(pad, offset=n * row)

    try:
        firsts = list(range(0, n_instances, tile))
        if firsts:
            upload(0, firsts[0])
        for k, first in enumerate(firsts):
            n = min(tile, n_instances - first)
            nxt = None
            if k + 1 < len(firsts):
                nxt = threading.Thread(target=upload, args=((k + 1) & 1, firsts[k + 1]))
                nxt.start()
            batch.set_initial_witness_device(stage[k & 1].ptr)
            batch.solve()
            if nxt is not None:
                nxt.join()
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
                        e = i
                        while e < n and res[e].status == 0:
                            e += 1
                        out[first + i:first + e] = batch.extract(keep, i, e - i)
                        i = e
    finally:
        batch.free()
        for d in stage:
            d.free()
    return results, out
