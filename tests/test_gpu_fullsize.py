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


def test_config3_full_size_properties(oracle):
    """BASELINE config 3 at its own batch: sha256 -> keccak256 + 96 RANGE(8) over 2^16 instances. (1) digests of a sample against hashlib
    (SHA-256) and the oracle's Keccak; (2) batch-size independence against a 64-instance batch that IS compared with the oracle bit for bit;
    (3) determinism over two solves (per-instance digest of the whole map); (4) the failing instances are exactly the planted ones (an
    input byte that is no byte: RANGE fails at that input's opcode); (5) the same per-instance digests through acvm_node_solve in tiles."""
    import hashlib
    import ctypes as C
    B = 1 << 16
    circ, ids = synth.hash_circuit()
    data = circ.to_bytes()
    vals = np.frombuffer(synth.byte_batch(B, len(ids)), dtype=np.uint8).reshape(B, len(ids), 32).copy()
    planted = {5: 0, 4097: 95, 30000: 64, B - 1: 17}  # instance -> input whose value is 256 + something
    for j, k in planted.items():
        vals[j, k, 30] = 1
    values = vals.tobytes()
    gc = acvm_amd.Circuit(data)
    batch = acvm_amd.Batch(gc, B, ids)
    batch.set_initial_witness(values)
    n_bad = batch.solve()
    res = batch.results()
    failed = {j: res[j] for j in range(B) if res[j].status != acvm_amd.STATUS_SOLVED}
    assert n_bad == len(failed) == len(planted) and set(failed) == set(planted)
    for j, k in planted.items():  # RANGE opcode k is the check of input k (synth.hash_circuit)
        assert (failed[j].err, failed[j].opcode_index) == (acvm_amd.ERR_UNSATISFIED, k), j
    sample = [0, 1, 63, 64, 4096, 12345, 40000, B - 2]
    out = batch.extract(circ.return_values, 0, 1)  # (a solved instance has all 32 Keccak outputs)
    assert out.shape == (1, 32, 32)
    sha_out, kec_out = list(range(len(ids) + 1, len(ids) + 33)), list(range(len(ids) + 33, len(ids) + 65))
    for j in sample:
        msg = bytes(vals[j, :64, 31])
        d1 = hashlib.sha256(msg).digest()
        got1 = bytes(batch.extract(sha_out, j, 1)[0, :, 31])
        assert got1 == d1, j
        buf = C.create_string_buffer(32)
        m2 = d1 + bytes(vals[j, 64:, 31])
        oracle.lib().oracle_keccak256(m2, len(m2), buf)
        assert bytes(batch.extract(kec_out, j, 1)[0, :, 31]) == buf.raw, j
    # (2) the small batch against the oracle, the big one against the small one
    picks = sample + [5, 4097]
    sub = b"".join(values[j * len(ids) * 32:(j + 1) * len(ids) * 32] for j in picks)
    small = acvm_amd.Batch(gc, len(picks), ids)
    small.set_initial_witness(sub)
    small.solve()
    sres, (sasg, svals) = small.results(), small.witness_map()
    ores, oasg, ovals = oracle.solve_batch(oracle.Circuit(data), ids, sub, len(picks))
    for i, j in enumerate(picks):
        assert sres[i].as_tuple() == ores[i].as_tuple() == res[j].as_tuple(), j
        assert np.array_equal(sasg[i], oasg[i][: sasg.shape[1]]) and np.array_equal(svals[i], ovals[i][: svals.shape[1]])
        asg, v = batch.witness_map(j, 1)
        assert np.array_equal(asg[0], sasg[i]) and np.array_equal(v[0], svals[i]), j
    # (3) determinism, (5) the node driver in tiles: one digest per instance, three ways
    dig = batch.digest()
    batch.reset()
    assert batch.solve() == n_bad and np.array_equal(batch.digest(), dig)
    for i, j in enumerate(picks):
        assert bytes(dig[j]) == oracle.witness_map_digest(oasg[i], ovals[i]), j
    small.free()
    batch.free()
    node = acvm_amd.Node(gc, ids, keep=circ.return_values, devices=[0], tile=1 << 14)
    not_solved, nres, kept, asg, ndig = node.solve(values, B)
    node.free()
    assert not_solved == n_bad and np.array_equal(ndig, dig)
    assert [r.as_tuple() for r in nres] == [r.as_tuple() for r in res]
    assert not kept[12345, :, :31].any() and asg[12345].all()  # (the kept witnesses are the Keccak digest bytes)


def test_config4_full_size_properties(oracle):
    """BASELINE config 4 at its own batch: Pedersen + FixedBaseScalarMul + SchnorrVerify over 2^16 instances whose rows repeat a 1 024-row
    pattern (rows 0..7 of it break the limb / modulus checks, odd rows carry a flipped signature bit). The 1 024-row batch is compared with the
    oracle bit for bit; every instance of the big batch must equal its row of the pattern (results, return witnesses, digest of the map);
    the instances that fail are exactly the planted ones; two solves agree; acvm_node_solve in tiles returns the same digests."""
    B, PAT = 1 << 16, 1024
    circ, ids = synth.grumpkin_circuit()
    data = circ.to_bytes()
    base = synth.grumpkin_rows(PAT, first_instance=0)
    arr = np.frombuffer(synth.values_from_rows(base), dtype=np.uint8).reshape(PAT, -1)
    values = arr[np.arange(B) % PAT].tobytes()
    gc = acvm_amd.Circuit(data)
    ret = circ.return_values
    small = acvm_amd.Batch(gc, PAT, ids)
    small.set_initial_witness(arr.tobytes())
    small.solve()
    sres, sdig = small.results(), small.digest()
    ores, oasg, ovals = oracle.solve_batch(oracle.Circuit(data), ids, arr.tobytes(), PAT, n_threads=8)
    sasg, svals = small.witness_map()
    assert [r.as_tuple() for r in sres] == [r.as_tuple() for r in ores]
    assert np.array_equal(sasg, oasg[:, : sasg.shape[1]]) and np.array_equal(svals, ovals[:, : svals.shape[1]])
    failing = [j for j in range(PAT) if sres[j].status != acvm_amd.STATUS_SOLVED]
    # the planted failures: the limb / modulus rows (scalar_mul.rs:25-51); a flipped signature is a Solved instance whose verdict witness is 0
    assert failing and set(failing) <= set(range(8)) and all(sres[j].err == acvm_amd.ERR_BLACKBOX_FAILED for j in failing)
    verdict = ret[-1]
    ok = [int(svals[j, verdict, 31]) for j in range(PAT) if j not in failing]
    assert 0 < sum(ok) < len(ok)  # both verdicts occur
    small.free()
    batch = acvm_amd.Batch(gc, B, ids)
    batch.set_initial_witness(values)
    n_bad = batch.solve()
    res = batch.results()
    assert n_bad == len(failing) * (B // PAT)
    assert all(res[j].as_tuple() == sres[j % PAT].as_tuple() for j in range(B))
    dig = batch.digest()
    assert np.array_equal(dig, np.tile(sdig, (B // PAT, 1)))  # the whole map of every instance == its row of the pattern
    good = [j for j in range(PAT) if j not in failing]
    kept_small = np.stack([svals[j][ret] for j in good])
    for rep in (0, 17, B // PAT - 1):
        first = rep * PAT
        for j in good[:40]:
            assert np.array_equal(batch.extract(ret, first + j, 1)[0], svals[j][ret]), (rep, j)
    assert kept_small.shape == (len(good), len(ret), 32)
    batch.reset()
    assert batch.solve() == n_bad and np.array_equal(batch.digest(), dig)
    batch.free()
    node = acvm_amd.Node(gc, ids, keep=ret, devices=[0], tile=1 << 14)
    not_solved, nres, kept, asg, ndig = node.solve(values, B)
    node.free()
    assert not_solved == n_bad and np.array_equal(ndig, dig) and [r.as_tuple() for r in nres] == [r.as_tuple() for r in res]
    for j in good[:40]:
        assert asg[PAT * 3 + j].all() and np.array_equal(kept[PAT * 3 + j], svals[j][ret])


def test_config2_at_the_metric_batch_through_the_node_driver():
    """The metric's configuration end to end: 10k gates x 2^20 instances from HOST memory through acvm_node_solve (tiles of 2^17 on the one
    device, uploads beside the solves, the exact path of the edge-case instances beside the next tile). The per-instance digests -- hence the
    digest of digests bench.py prints -- equal those of the tiled batch API over resident inputs."""
    from acvm_amd import shard, tiling
    total, tile = 1 << 20, 1 << 17
    circ, ids = synth.arithmetic_circuit(10000, seed=0xAC1D0002)
    gc = acvm_amd.Circuit(circ.to_bytes())
    values = synth.witness_batch(total, seed=0xAC1D0002)
    sh = tiling.ResidentShard(gc, ids, values, total, tile)
    dig = sh.digests()
    sh.free()
    node = acvm_amd.Node(gc, ids, keep=gc.witness_set("return_values"), devices=[0], tile=tile)
    not_solved, res, kept, asg, ndig = node.solve(values, total, results=True)
    st = node.stats()
    node.free()
    assert np.array_equal(ndig, dig)
    assert shard.digest_of_digests(shard.chunk_digests(ndig)) == shard.digest_of_digests(shard.chunk_digests(dig))
    failed = [j for j in range(total) if res[j].status != acvm_amd.STATUS_SOLVED]
    assert not_solved == len(failed) and set(failed) <= set(range(8))
    assert st["tiles"] == [total // tile] and asg[8:].all()


def test_solve_then_import_equals_plain_tiles(oracle):
    """acvm_batch_solve_then_import: the next tile's import rides behind the solve, gated on the device by "no instance left the generic path".
    Six tiles of 512 instances through one handle (tile 0 holds the edge-case inputs: its successor's import is held back and happens at the
    set_initial_witness call), two passes (the last tile prefetches tile 0 of the next pass): results and return witnesses equal those of the
    plain load / solve loop and the oracle's; an initial witness, a whole map or a digest cannot be read while the next tile sits in the table."""
    from acvm_amd import tiling
    circ, ids = synth.mixed_circuit(800, seed=0x7111E)
    data = circ.to_bytes()
    n, tile = 3072, 512
    values = synth.witness_batch(n, seed=0x7111E, edge_cases=True)
    gc = acvm_amd.Circuit(data)
    ret = gc.witness_set("return_values")
    plain = tiling.ResidentShard(gc, ids, values, n, tile)
    want = []
    for k in range(len(plain.starts)):
        plain.load_tile(k)
        plain.batch.solve()
        res = plain.batch.results()
        want.append(([r.as_tuple() for r in res], [plain.batch.witness(w) for w in ret], plain.batch.digest()))
    plain.free()
    sh = tiling.ResidentShard(gc, ids, values, n, tile)
    held_back = 0
    for rep in range(2):
        for k in range(len(sh.starts)):
            sh.load_tile(k)
            sh.solve_tile(k)
            res = [r.as_tuple() for r in sh.batch.results()]
            assert res == want[k][0], (rep, k)
            flagged = sh.batch.stats()["n_slow_instances"]
            held_back += flagged > 0
            for w, (v, a) in zip(ret, want[k][1]):
                gv, ga = sh.batch.witness(w)
                assert np.array_equal(gv, v) and np.array_equal(ga, a), (rep, k, w)
            if flagged == 0:  # the next tile's inputs are in the table already
                for call in (lambda: sh.batch.witness(ids[0]), lambda: sh.batch.witness_map(0, 1), lambda: sh.batch.digest()):
                    with pytest.raises(acvm_amd.AcvmError, match="initial witnesses of this solve are gone"):
                        call()
            else:  # held back: everything of this tile is still there
                assert np.array_equal(sh.batch.digest(), want[k][2])
    assert held_back == 2  # tile 0 of either pass
    # ... and a plain solve afterwards sees its own tile whole
    sh.load_tile(3)
    sh.batch.solve()
    assert np.array_equal(sh.batch.digest(), want[3][2])
    sh.free()
    ores, oasg, ovals = oracle.solve_batch(oracle.Circuit(data), ids, values[: 16 * len(ids) * 32], 16)
    assert [r.as_tuple() for r in ores] == want[0][0][:16]
