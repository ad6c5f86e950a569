"""BASELINE config 5 at circuit size in the driver-run suite, with and without witness-slot liveness reuse: the 10^6-opcode mixed circuit of acvm_amd.synth
(16 memory blocks x 64 cells with per-instance dynamic indices, ToLeRadix(256, 4 limbs), every opcode class), ONE tile of 4 096
instances through the level kernels, per-instance digests of the witness maps, and an audit sample re-solved
by the CPU oracle: status tuples, return witnesses and digests (hashlib over the oracle's full map) bit for bit."""
import os

import numpy as np
import pytest

import acvm_amd
from acvm_amd import synth

pytestmark = pytest.mark.gpu


def test_million_opcode_tile_against_oracle_audit(oracle):
    G, tile = 1_000_000, 4096
    circ, ids = synth.mixed_circuit(G)
    data = circ.to_bytes()
    gc = acvm_amd.Circuit(data)
    assert gc.num_opcodes >= G
    ret = gc.witness_set("return_values")
    batch = acvm_amd.Batch(gc, tile, ids)
    st0 = batch.stats()
    assert st0["truncated_at"] == 0xFFFFFFFF  # the whole circuit is on the level path
    values = synth.witness_batch(tile, seed=0xAC1D0005)
    batch.set_initial_witness(values)
    n_bad = batch.solve()
    res = batch.results()
    assert n_bad == sum(1 for r in res if r.status != 0) <= 8  # only the edge-case inputs of the synthetic batch may fail
    dig = batch.digest()
    # a second solve of the same tile through the reused handle: same digests
    batch.reset()
    batch.solve()
    assert np.array_equal(dig, batch.digest())
    # 32 instances (the edge cases 0 and 5 among them), one oracle thread each: ~1.4 s of a core per instance
    picks = sorted(set([0, 5, 9, 1000, tile - 1] + [int(x) for x in np.linspace(10, tile - 2, 27)]))
    assert len(picks) >= 32
    row = len(ids) * 32
    sub = b"".join(values[j * row:(j + 1) * row] for j in picks)
    ores, oasg, ovals = oracle.solve_batch(oracle.Circuit(data), ids, sub, len(picks), n_threads=min(len(picks), os.cpu_count() or 1))
    for i, j in enumerate(picks):
        assert res[j].as_tuple() == ores[i].as_tuple(), j
        assert bytes(dig[j]) == oracle.witness_map_digest(oasg[i], ovals[i]), j
        if ores[i].status == 0:
            got = batch.extract(ret, j, 1)[0]
            assert all(bytes(got[n]) == bytes(ovals[i][w]) for n, w in enumerate(ret)), j
    ret_plain = batch.extract(ret, 8, tile - 8)
    batch.free()
    # The same circuit with witness-slot liveness reuse: about half the rows, so a tile of TWICE the instances in the
    # same memory; digests folded into the solve. Its first 4 096 instances are the plain tile's: same results, digests, return values
    # (the edge-case instances 0..7 are re-solved from their initial witnesses in the exact path's own table).
    big = 2 * tile
    reuse = acvm_amd.Batch(gc, big, ids, reuse_slots=True, keep=ret)
    st = reuse.stats()
    assert st["n_table_rows"] < 0.6 * st["n_witnesses"]
    reuse.set_initial_witness(synth.witness_batch(big, seed=0xAC1D0005))
    assert reuse.solve() == n_bad
    res2 = reuse.results()
    assert [r.as_tuple() for r in res2[:tile]] == [r.as_tuple() for r in res]
    assert np.array_equal(reuse.digest(0, tile), dig)
    assert np.array_equal(reuse.extract(ret, 8, tile - 8), ret_plain)
    assert all(res2[j].status == 0 for j in range(tile, big))
    reuse.free()
