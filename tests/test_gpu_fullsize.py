"""Full-size runs (BASELINE config 2: 10k gates x 2^16 instances) checked through size-independent properties, since the
CPU oracle cannot solve 65 536 instances in a test: (1) every solved instance satisfies every opcode of the circuit,
re-evaluated here with Python big integers (independent of oracle and kernels) on a random sample; (2) an instance solved
inside the big batch is bit-identical to the same instance solved in a small batch (no cross-instance coupling, no
dependence on the batch size); (3) solving twice gives the same witness table (determinism); (4) the instances that fail
are exactly the edge-case inputs."""
import random

import numpy as np
import pytest

import acvm_amd
from acvm_amd import synth
from acvm_amd.acir import P

pytestmark = pytest.mark.gpu


def check_arithmetic_satisfied(circ, wmap):
    for k, e in enumerate(circ.opcodes):
        acc = e.q_c
        for c, l, r in e.mul_terms:
            acc += c * wmap[l] * wmap[r]
        for c, w in e.linear_combinations:
            acc += c * wmap[w]
        assert acc % P == 0, f"opcode {k} not satisfied"


def test_config2_full_size_properties():
    B = 1 << 16
    circ, ids = synth.arithmetic_circuit(10000, seed=0xAC1D0002)
    data = circ.to_bytes()
    values = synth.witness_batch(B, seed=0xAC1D0002)
    batch = acvm_amd.Batch(acvm_amd.Circuit(data), B, ids)
    batch.set_initial_witness(values)
    n_bad = batch.solve()
    res = batch.results()
    failed = [j for j in range(B) if res[j].status != acvm_amd.STATUS_SOLVED]
    assert n_bad == len(failed) and set(failed) <= set(range(8)), failed[:16]  # only the edge-case inputs may fail
    r = random.Random(7)
    sample = sorted(r.sample(range(8, B), 24) + [8, B - 1])
    # (1) constraint satisfaction, independent big-integer evaluation
    for j in sample[:12]:
        asg, vals = batch.witness_map(j, 1)
        assert asg[0, 1:].all()
        wmap = {w: int.from_bytes(vals[0, w].tobytes(), "big") for w in range(asg.shape[1])}
        for k, w in enumerate(ids):
            assert wmap[w] == int.from_bytes(values[(j * len(ids) + k) * 32:(j * len(ids) + k + 1) * 32], "big") % P
        check_arithmetic_satisfied(circ, wmap)
    # (2) batch-size independence: the same instances solved as a batch of 26
    small_vals = b"".join(values[j * len(ids) * 32:(j + 1) * len(ids) * 32] for j in sample)
    small = acvm_amd.Batch(acvm_amd.Circuit(data), len(sample), ids)
    small.set_initial_witness(small_vals)
    assert small.solve() == 0
    sasg, svals = small.witness_map()
    for i, j in enumerate(sample):
        asg, vals = batch.witness_map(j, 1)
        assert np.array_equal(asg[0], sasg[i]) and np.array_equal(vals[0], svals[i]), j
    # (3) determinism: a second solve leaves every return witness unchanged
    ret = circ.return_values[0]
    v1, a1 = batch.witness(ret)
    batch.reset()
    assert batch.solve() == n_bad
    v2, a2 = batch.witness(ret)
    assert np.array_equal(v1, v2) and np.array_equal(a1, a2)
    # (5) a checksum of checksums over EVERY instance: the per-instance digest of the whole witness map (acvm_batch_digest) is the
    # same when the 65 536 instances are solved as eight tiles of 8 192 through one reused handle -- (2) for all instances, not a
    # sample -- and equals the digest computed with hashlib over the CPU oracle's map for the sampled instances
    from acvm_amd.tiling import solve_tiled
    from oracle import binding as oracle
    dig = batch.digest()
    small.free()
    batch.free()
    dig_t = np.zeros((B, 32), dtype=np.uint8)
    res_t, _ = solve_tiled(acvm_amd.Circuit(data), ids, values, B, 8192, [], digests=dig_t)
    assert [r.as_tuple() for r in res_t] == [r.as_tuple() for r in res]
    assert np.array_equal(dig, dig_t)
    assert len({bytes(d) for d in dig[8:]}) == B - 8                      # distinct inputs, distinct maps
    picks = sample[:6] + [0, 3]
    sub = b"".join(values[j * len(ids) * 32:(j + 1) * len(ids) * 32] for j in picks)
    ores, oasg, ovals = oracle.solve_batch(oracle.Circuit(data), ids, sub, len(picks))
    for i, j in enumerate(picks):
        assert ores[i].as_tuple() == res[j].as_tuple()
        assert bytes(dig[j]) == oracle.witness_map_digest(oasg[i], ovals[i]), j


def test_mixed_circuit_full_batch_properties():
    """Config-5 opcode mix at batch 2^14: batch-size independence against a small batch (which tests/test_gpu_opcodes.py
    pins against the oracle) and determinism."""
    B = 1 << 14
    circ, ids = synth.mixed_circuit(2500)
    data = circ.to_bytes()
    values = synth.witness_batch(B, seed=0xAC1D0005)
    batch = acvm_amd.Batch(acvm_amd.Circuit(data), B, ids)
    batch.set_initial_witness(values)
    n_bad = batch.solve()
    assert n_bad <= 8
    sample = [0, 3, 9, 100, 4097, B - 1]
    small = acvm_amd.Batch(acvm_amd.Circuit(data), len(sample), ids)
    small.set_initial_witness(b"".join(values[j * len(ids) * 32:(j + 1) * len(ids) * 32] for j in sample))
    small.solve()
    sres = small.results()
    res = batch.results()
    sasg, svals = small.witness_map()
    for i, j in enumerate(sample):
        assert res[j].as_tuple() == sres[i].as_tuple(), j
        asg, vals = batch.witness_map(j, 1)
        assert np.array_equal(asg[0], sasg[i]) and np.array_equal(vals[0], svals[i]), j


def test_hundred_thousand_gate_circuit(oracle):
    """Towards config 5's circuit size: 100 000 gates x 2 048 instances (6.5 GB witness table, 52 levels of up to 7 000
    gates). Four instances are checked bit for bit against the CPU oracle, a sample re-evaluates every constraint with
    Python integers, and only the edge-case inputs may fail."""
    B, G = 2048, 100000
    circ, ids = synth.arithmetic_circuit(G, seed=0xAC1D0005)
    data = circ.to_bytes()
    values = synth.witness_batch(B, seed=0xAC1D0005)
    gc = acvm_amd.Circuit(data)
    batch = acvm_amd.Batch(gc, B, ids)
    batch.set_initial_witness(values)
    n_bad = batch.solve()
    res = batch.results()
    failed = [j for j in range(B) if res[j].status != acvm_amd.STATUS_SOLVED]
    assert n_bad == len(failed) and set(failed) <= set(range(8)), failed[:16]
    st = batch.stats()
    assert st["n_opcodes"] == G and st["n_fast_gates"] + st["n_dyn_gates"] == G and st["truncated_at"] == 0xFFFFFFFF
    picks = [0, 5, 8, B - 1]
    sub = b"".join(values[j * len(ids) * 32:(j + 1) * len(ids) * 32] for j in picks)
    ores, oasg, ovals = oracle.solve_batch(oracle.Circuit(data), ids, sub, len(picks))
    for i, j in enumerate(picks):
        assert res[j].as_tuple() == ores[i].as_tuple(), j
        asg, vals = batch.witness_map(j, 1)
        nw = min(asg.shape[1], oasg.shape[1])
        assert np.array_equal(asg[0, :nw], oasg[i, :nw]) and np.array_equal(vals[0, :nw], ovals[i, :nw]), j
    asg, vals = batch.witness_map(777, 1)
    check_arithmetic_satisfied(circ, {w: int.from_bytes(vals[0, w].tobytes(), "big") for w in range(asg.shape[1])})
    batch.free()


def test_tiled_solve_matches_one_batch():
    """Tiles inside one device (acvm_amd/tiling.py): 700 instances in tiles of 256 (the last one partial) give the
    same per-instance results and return witnesses as one batch of 700, failing edge-case instances included."""
    from acvm_amd.tiling import solve_tiled
    circ, ids = synth.mixed_circuit(1200)
    data = circ.to_bytes()
    B = 700
    values = synth.witness_batch(B, seed=0xAC1D0005)
    gc = acvm_amd.Circuit(data)
    ret = gc.witness_set("return_values")
    assert ret
    dig_t = np.zeros((B, 32), dtype=np.uint8)
    res_t, vals_t = solve_tiled(gc, ids, values, B, 256, ret, digests=dig_t)
    batch = acvm_amd.Batch(gc, B, ids)
    batch.set_initial_witness(values)
    batch.solve()
    res = batch.results()
    assert np.array_equal(dig_t, batch.digest())
    assert [r.as_tuple() for r in res_t] == [r.as_tuple() for r in res]
    for j in range(B):
        if res[j].status == 0:
            assert np.array_equal(vals_t[j], batch.extract(ret, j, 1)[0]), j
        else:
            assert not vals_t[j].any()
    assert sum(1 for r in res if r.status != 0) <= 8
    batch.free()
