"""GPU parity of every non-arithmetic opcode kind against the CPU oracle (bit-exact: status, error kind, failing
opcode, aux values, assigned set, every witness value), through the level kernels AND through the exact in-order
kernels (force_slow), including the failure / edge cases the reference tests exercise."""
import hashlib
import random

import numpy as np
import pytest

from acvm_amd.acir import (P, BlackBoxFuncCall as BB, Brillig, Circuit, Expression as E, FunctionInput as FI, MemoryInit,
                           MemoryOp, QuotientDirective, ToLeRadix)
from acvm_amd.synth import values_from_rows

pytestmark = pytest.mark.gpu


def run_both(oracle, circ, ids, rows, force_slow=False):
    import acvm_amd
    B = len(rows)
    values = values_from_rows(rows)
    data = circ.to_bytes()
    ores, oasg, ovals = oracle.solve_batch(oracle.Circuit(data), ids, values, B)
    batch = acvm_amd.Batch(acvm_amd.Circuit(data), B, ids)
    batch.set_force_slow_path(force_slow)
    batch.set_initial_witness(values)
    batch.solve()
    gres = batch.results()
    gasg, gvals = batch.witness_map()
    stats = batch.stats()
    batch.free()
    for j in range(B):
        assert gres[j].as_tuple() == ores[j].as_tuple(), f"instance {j}: gpu {gres[j].as_tuple()} oracle {ores[j].as_tuple()} (slow={force_slow})"
        if ores[j].message:
            assert gres[j].message == ores[j].message, (j, gres[j].message, ores[j].message)
    nw = min(oasg.shape[1], gasg.shape[1])
    assert np.array_equal(oasg[:, :nw], gasg[:, :nw]), f"assigned sets differ (slow={force_slow})"
    bad = np.argwhere((ovals[:, :nw] != gvals[:, :nw]).any(axis=2))
    assert bad.size == 0, f"witness values differ at (instance, witness) {bad[:8].tolist()} (slow={force_slow})"
    return ores, stats


def both_paths(oracle, circ, ids, rows):
    ores, stats = run_both(oracle, circ, ids, rows, force_slow=False)
    run_both(oracle, circ, ids, rows, force_slow=True)
    return ores, stats


def rnd(seed):
    return random.Random(seed)


def test_range_and_logic(oracle):
    r = rnd(1)
    ops = [BB("RANGE", {"input": FI(1, 8)}), BB("RANGE", {"input": FI(2, 64)}), BB("RANGE", {"input": FI(3, 254)}),
           BB("AND", {"lhs": FI(1, 8), "rhs": FI(2, 8), "output": 5}), BB("XOR", {"lhs": FI(2, 64), "rhs": FI(3, 64), "output": 6}),
           BB("XOR", {"lhs": FI(3, 254), "rhs": FI(4, 254), "output": 7}), BB("AND", {"lhs": FI(3, 256), "rhs": FI(4, 256), "output": 8}),
           BB("AND", {"lhs": FI(3, 13), "rhs": FI(4, 13), "output": 9}), BB("XOR", {"lhs": FI(4, 0), "rhs": FI(3, 0), "output": 10})]
    circ = Circuit(10, ops)
    rows = []
    for j in range(96):
        a, b = r.randrange(256), r.randrange(1 << 64)
        rows.append([a, b, r.randrange(P), r.randrange(P)])
    rows[3][0] = 256            # RANGE(8) fails at opcode 0
    rows[4][1] = 1 << 64        # RANGE(64) fails at opcode 1
    rows[5][2] = P - 1          # 254 bits: passes
    rows[6] = [255, (1 << 64) - 1, P - 1, P - 1]
    rows[7] = [0, 0, 0, 0]
    ores, _ = both_paths(oracle, circ, [1, 2, 3, 4], rows)
    assert ores[3].status == 2 and ores[3].opcode_index == 0 and ores[4].opcode_index == 1 and ores[5].status == 0


def test_small_range_edges(oracle):
    """RANGE with 0..8 bits takes the low-limb check (ops_light.hpp op_range): values whose low bits are in range but whose upper bits are not,
    the multiples of 2^29 and 2^32 around the limb boundaries, p - 1, and the exact boundary 2^bits - 1 / 2^bits for every width."""
    ops = [BB("RANGE", {"input": FI(1 + b, b)}) for b in range(9)]
    circ = Circuit(9, ops)
    ids = list(range(1, 10))
    rows = []
    for b in range(9):                       # boundary of width b at witness 1 + b; the other witnesses zero
        for v in ((1 << b) - 1, 1 << b):
            row = [0] * 9
            row[b] = v
            rows.append(row)
    for hi in (1 << 29, 1 << 32, 1 << 58, 1 << 64, 1 << 253, P - 1, P - 256, (1 << 29) + 5, (1 << 200) + 255):
        rows.append([0] * 8 + [hi])                  # low bits of RANGE(8)'s input in range, the value is not
        rows.append([hi & 0] * 8 + [hi & 0xff])      # and the byte alone: passes
    ores, _ = both_paths(oracle, circ, ids, rows)
    for b in range(9):
        assert ores[2 * b].status == 0 and ores[2 * b + 1].status == 2 and ores[2 * b + 1].opcode_index == b


def test_logic_bit_mismatch_panics(oracle):
    circ = Circuit(3, [BB("AND", {"lhs": FI(1, 8), "rhs": FI(2, 16), "output": 3})])
    ores, _ = both_paths(oracle, circ, [1, 2], [[1, 2]] * 3)
    assert ores[0].err == oracle.E_PANIC


def test_output_conflict_and_missing_input(oracle):
    # witness 3 is an input AND the output of XOR: insert_value conflict unless equal (pwg/mod.rs:338-357)
    circ = Circuit(4, [BB("XOR", {"lhs": FI(1, 8), "rhs": FI(2, 8), "output": 3}), BB("RANGE", {"input": FI(4, 8)})])
    rows = [[5, 3, 6], [5, 3, 7], [0, 0, 0]]
    ores, _ = both_paths(oracle, circ, [1, 2, 3], rows)
    assert ores[0].err == oracle.E_MISSING_ASSIGNMENT and ores[0].opcode_index == 1  # 5^3 == 6 passes, then w4 missing
    assert ores[1].err == oracle.E_UNSATISFIED and ores[1].opcode_index == 0


def test_directives(oracle):
    r = rnd(2)
    ops = [QuotientDirective(E.from_witness(1), E.from_witness(2), 4, 5),
           QuotientDirective(E([(1, 1, 2)], [(3, 1)], 7), E([], [(1, 2), (1, 3)], 0), 6, 7, predicate=E.from_witness(3)),
           ToLeRadix(E.from_witness(1), list(range(8, 8 + 32)), 256),
           ToLeRadix(E.from_witness(2), list(range(40, 40 + 8)), 2),
           ToLeRadix(E([], [(1, 2), (1, 3)], 5), list(range(48, 48 + 6)), 10),
           ToLeRadix(E.from_witness(3), list(range(54, 54 + 3)), 16)]
    circ = Circuit(57, ops)
    rows = []
    for j in range(64):
        rows.append([r.randrange(P), r.randrange(256), r.randrange(2) * r.randrange(1 << 12)])
    rows[0] = [P - 1, 0, 0]           # division by zero -> (0, 0)
    rows[1] = [12345, 7, 1]
    rows[2] = [0, 255, 4095]
    rows[3] = [P - 1, 256, 1]         # radix 2 with 9 bits > 8 outputs -> Unsatisfied
    rows[4] = [5, 200, 999999]        # radix 10 needs 7 digits > 6 outputs; radix 16 too
    rows[5] = [1 << 200, 1, 4096]     # radix 16 needs 4 digits > 3
    both_paths(oracle, circ, [1, 2, 3], rows)


def test_memory_ops(oracle):
    r = rnd(3)
    n = 6
    init_ws = list(range(1, n + 1))                 # inputs 1..6 = cells, 7 = index, 8 = predicate, 9 = value
    ops = [MemoryInit(0, init_ws),
           MemoryOp(0, E.constant(0), E.from_witness(7), E.from_witness(10)),                       # read  w10 = m[w7]
           MemoryOp(0, E.constant(1), E([], [(1, 7)], 1), E.from_witness(9), predicate=E.from_witness(8)),  # m[w7+1] = w9 if w8
           MemoryOp(0, E.constant(0), E([], [(1, 7)], 1), E.from_witness(11)),                      # read  w11 = m[w7+1]
           MemoryOp(0, E.constant(0), E.constant(2), E.from_witness(12), predicate=E.from_witness(8)),  # w12 = m[2] or 0
           MemoryInit(1, [10, 11]),
           MemoryOp(1, E.constant(1), E.constant(0), E([(1, 10, 11)], [(2, 12)], 3)),               # m1[0] = w10*w11+2*w12+3
           MemoryOp(1, E.constant(0), E.constant(0), E.from_witness(13)),
           E([(1, 13, 13)], [(-1 % P, 14)], 0)]                                                      # w14 = w13^2
    circ = Circuit(14, ops)
    rows = []
    for j in range(80):
        rows.append([r.randrange(P) for _ in range(n)] + [r.randrange(n - 1), r.randrange(2), r.randrange(P)])
    rows[0][6] = n - 1      # read ok, write index n -> IndexOutOfBounds when predicate set
    rows[0][7] = 1
    rows[1][6] = n - 1      # predicate 0: write skipped, the following read at n fails
    rows[1][7] = 0
    rows[2][6] = n + 3      # first read out of bounds
    rows[3][6] = 1 << 70    # index does not fit u64 -> panic
    rows[4][6] = (1 << 32) + 1  # `as u32` wraps to 1
    ores, stats = both_paths(oracle, circ, list(range(1, 10)), rows)
    assert ores[0].err == oracle.E_INDEX_OOB and ores[2].err == oracle.E_INDEX_OOB and ores[3].err == oracle.E_PANIC


def test_memory_exact_replay_after_later_write(oracle):
    """An instance that fails AFTER a memory write at an EARLIER level must see the pre-write cell again when the exact
    path resumes at its event: the event opcode sits before the read in program order."""
    ops = [MemoryInit(0, [1, 2]),
           BB("RANGE", {"input": FI(3, 8)}),                                       # event for instances with w3 >= 256
           MemoryOp(0, E.constant(0), E.constant(0), E.from_witness(5)),          # w5 = m[0]
           MemoryOp(0, E.constant(1), E.constant(0), E.from_witness(4)),          # m[0] = w4
           MemoryOp(0, E.constant(0), E.constant(0), E.from_witness(6))]          # w6 = m[0]
    circ = Circuit(6, ops)
    rows = [[11, 22, 5, 99], [11, 22, 300, 99], [1, 2, 255, 3]]
    both_paths(oracle, circ, [1, 2, 3, 4], rows)


def test_recursive_aggregation_zero_fill(oracle):
    circ = Circuit(8, [BB("RecursiveAggregation", {"verification_key": [FI(1, 254)], "proof": [FI(2, 254)], "public_inputs": [FI(3, 254)],
                                                   "key_hash": FI(4, 254), "input_aggregation_object": None,
                                                   "output_aggregation_object": [5, 6, 7, 8]})])
    both_paths(oracle, circ, [1, 2, 3, 4], [[1, 2, 3, 4], [0, 0, 0, 0]])


def _hash_rows(r, B, n):
    return [[r.randrange(256) for _ in range(n)] for _ in range(B)]


@pytest.mark.parametrize("name,n", [("SHA256", 0), ("SHA256", 3), ("SHA256", 55), ("SHA256", 56), ("SHA256", 64), ("SHA256", 119), ("SHA256", 200),
                                    ("Blake2s", 0), ("Blake2s", 1), ("Blake2s", 64), ("Blake2s", 65), ("Blake2s", 130),
                                    ("Keccak256", 0), ("Keccak256", 1), ("Keccak256", 135), ("Keccak256", 136), ("Keccak256", 137), ("Keccak256", 300)])
def test_hash_opcodes(oracle, name, n):
    r = rnd(100 + n)
    ids = list(range(1, n + 1))
    outs = list(range(n + 1, n + 33))
    circ = Circuit(n + 32, [BB(name, {"inputs": [FI(w, 8) for w in ids], "outputs": outs})])
    rows = _hash_rows(r, 70, n)
    ores, _ = both_paths(oracle, circ, ids, rows)
    # independent anchor: hashlib for sha256 / blake2s
    if name in ("SHA256", "Blake2s") and n:
        import acvm_amd
        data = circ.to_bytes()
        batch = acvm_amd.Batch(acvm_amd.Circuit(data), len(rows), ids)
        batch.set_initial_witness(values_from_rows(rows))
        batch.solve()
        _, vals = batch.witness_map()
        h = hashlib.sha256 if name == "SHA256" else hashlib.blake2s
        for j in (0, 33, 69):
            assert bytes(vals[j, n + 1:n + 33, 31]) == h(bytes(rows[j])).digest()


def test_hash_byte_messages_four_wave_kernel(oracle):
    """The level kernel of byte messages (kernels_hash.hip hash_coop_level_kernel): widths 1..8 bits with arbitrary field values behind them
    (fetch_nearest_bytes keeps the low byte whatever num_bits says), lengths around the word and block boundaries, an output that is also an
    input (insert_value compares), three hashes in one level, a batch that is not a multiple of 64."""
    r = rnd(77)
    n = 71
    ids = list(range(1, n + 1))
    widths = [1 + (k % 8) for k in range(n)]
    o = n + 1
    ops = [BB("SHA256", {"inputs": [FI(w, b) for w, b in zip(ids, widths)], "outputs": list(range(o, o + 32))}),
           BB("Keccak256", {"inputs": [FI(w, b) for w, b in zip(ids[:67], widths)], "outputs": list(range(o + 32, o + 64))}),
           BB("Blake2s", {"inputs": [FI(w, b) for w, b in zip(ids[:65], widths)], "outputs": [ids[70]] + list(range(o + 64, o + 95))}),
           BB("Keccak256", {"inputs": [FI(w, 8) for w in range(o, o + 64)], "outputs": list(range(o + 95, o + 127))})]  # second level
    circ = Circuit(o + 127, ops)
    rows = [[r.randrange(P) if r.random() < 0.5 else r.randrange(256) for _ in range(n)] for _ in range(131)]
    import hashlib as hl
    for j in (4, 9, 130):  # make the compared output right for some instances: Blake2s digest byte 0 == witness 71
        rows[j][70] = hl.blake2s(bytes(v & 0xff for v in rows[j][:65])).digest()[0]
    ores, stats = both_paths(oracle, circ, ids, rows)
    assert ores[4].status == 0 and ores[9].status == 0 and ores[130].status == 0
    assert sum(1 for x in ores if x.status == 2) >= 100  # the others fail at the Blake2s opcode (index 2) unless the byte happens to match
    assert all(x.opcode_index == 2 for x in ores if x.status == 2)


def test_hash_inputs_from_byte_planes(oracle):
    """Initial witnesses that byte-message hashes read carry a 4-byte copy written by the import (plan.hpp "Byte planes": low 29 bits of the canonical
    value + is-byte flag), which the hash kernel reads in place of the row: values around every boundary of that word (255 / 256, 2^29 - 1 / 2^29 /
    2^29 + 5, p - 1, 0), fused RANGE checks of 7 and 8 bits on them, a witness shared by two hashes and an arithmetic gate (which reads the ROW), the
    same circuit with the planes switched off, and the planner's counts."""
    import acvm_amd
    r = rnd(4242)
    n = 37
    ids = list(range(1, n + 1))
    o = n + 1
    ops = [BB("RANGE", {"input": FI(w, 8 if w % 3 else 7)}) for w in ids[:20]]
    ops += [BB("SHA256", {"inputs": [FI(w, 8) for w in ids], "outputs": list(range(o, o + 32))}),
            BB("Keccak256", {"inputs": [FI(w, 8) for w in ids[5:30]], "outputs": list(range(o + 32, o + 64))}),
            E([(1, 3, 4)], [(1, 5), (P - 1, o + 64)], 9)]
    circ = Circuit(o + 64, ops)
    edge = [0, 1, 127, 128, 255, 256, 257, (1 << 29) - 1, 1 << 29, (1 << 29) + 5, P - 1, P - 256, 1 << 200]
    rows = [[r.randrange(128 if (w <= 20 and w % 3 == 0) else 256) for w in ids] for _ in range(150)]  # (the 7-bit checks pass)
    for j in range(40, 150):  # one edge value somewhere in the message (most of them fail a RANGE check or only change the low byte)
        rows[j][r.randrange(n)] = edge[j % len(edge)]
    ores, _ = both_paths(oracle, circ, ids, rows)
    assert sum(1 for x in ores if x.status == 0) >= 40 and sum(1 for x in ores if x.status == 2) >= 20
    st = acvm_amd.Circuit(circ.to_bytes()).plan_stats(ids)
    assert st["n_byte_planes"] == n and st["n_byte_plane_reads"] == n + 25
    with acvm_amd.tuning(byte_plane=0):
        assert acvm_amd.Circuit(circ.to_bytes()).plan_stats(ids)["n_byte_planes"] == 0
        both_paths(oracle, circ, ids, rows)


@pytest.mark.parametrize("n", [1023, 1024, 1025])
def test_hash_byte_message_length_limit(oracle, n):
    """1024 bytes is the longest message the four-wave kernel takes (16 KiB of LDS per 64 instances); 1025 goes lane-per-instance."""
    r = rnd(n)
    ids = list(range(1, n + 1))
    circ = Circuit(n + 32, [BB("Keccak256", {"inputs": [FI(w, 8) for w in ids], "outputs": list(range(n + 1, n + 33))})])
    both_paths(oracle, circ, ids, _hash_rows(r, 67, n))


def test_hash_many_records_one_wave_instantiation(oracle):
    """more than 2 048 (record, 64 instances) groups in one launch: the one-wave instantiation of the same kernel"""
    r = rnd(5)
    n, n_rec = 24, 40
    ids = list(range(1, n + 1))
    ops = []
    for k in range(n_rec):
        name = ("SHA256", "Keccak256", "Blake2s")[k % 3]
        ops.append(BB(name, {"inputs": [FI(ids[(i + k) % n], 8) for i in range(5 + k % 19)], "outputs": list(range(n + 1 + 32 * k, n + 33 + 32 * k))}))
    circ = Circuit(n + 32 * n_rec, ops)
    B = 64 * 52 + 3  # 53 groups x 40 records = 2 120 groups
    ores, _ = run_both(oracle, circ, ids, _hash_rows(r, B, n))
    assert all(x.status == 0 for x in ores)


def test_range_checks_fused_into_hash_records(oracle):
    """Byte RANGE checks on the inputs of a byte-message hash run inside the hash's level kernel (plan.cpp): widths 0..8 (fused), 9 and 64
    (their own records), two checks on one witness, one witness in two input slots, checks on the outputs of the first hash that the second
    one reads, and values that break each of them -- the failing opcode must be the RANGE opcode, the lowest one when several fail."""
    r = rnd(31)
    n = 40
    ids = list(range(1, n + 1))
    o = n + 1
    ops = [BB("RANGE", {"input": FI(ids[k], k % 9)}) for k in range(18)]            # opcodes 0..17: widths 0..8 twice
    ops += [BB("RANGE", {"input": FI(ids[20], 9)}), BB("RANGE", {"input": FI(ids[21], 64)})]      # 18, 19: not byte-sized
    ops += [BB("RANGE", {"input": FI(ids[5], 3)})]                                  # 20: a second, narrower check on witness 6
    ops += [BB("SHA256", {"inputs": [FI(w, 8) for w in ids[:32]] + [FI(ids[3], 8)], "outputs": list(range(o, o + 32))})]          # 21
    ops += [BB("RANGE", {"input": FI(o + k, 8)}) for k in range(4)]                 # 22..25: on SHA outputs (always pass)
    ops += [BB("RANGE", {"input": FI(o + 4, 7)})]                                   # 26: fails for half of the instances
    ops += [BB("Keccak256", {"inputs": [FI(w, 8) for w in range(o, o + 32)] + [FI(w, 8) for w in ids[32:]], "outputs": list(range(o + 32, o + 64))})]  # 27
    ops += [BB("RANGE", {"input": FI(ids[k], 8)}) for k in range(32, 40)]           # 28..35: after the hash that reads them, in program order
    circ = Circuit(o + 64, ops)
    rows = []
    for j in range(150):
        row = [r.randrange(1 << (k % 9)) if k < 18 else r.randrange(256) for k in range(n)]
        row[5] = r.randrange(8)       # witness 6 also has the width-3 check of opcode 20
        rows.append(row)
    rows[1][0] = 1                      # RANGE(w1, 0) fails
    rows[2][8] = 256                    # RANGE(w9, 8): low byte in range, the value is not
    rows[3][8] = P - 1
    rows[4][13] = 16                    # width 4
    rows[5][5] = 31                     # passes width 5 (opcode 5), fails the second check of width 3 (opcode 20)
    rows[6][20] = 512                   # width 9, own record
    rows[7][35] = 1 << 29               # opcode 31, low limb zero
    rows[8][3] = 8; rows[8][12] = 8     # two failures: opcode 3 (width 3) before opcode 12
    rows[9][39] = 300                   # opcode 35
    ores, _ = both_paths(oracle, circ, ids, rows)
    want = {1: 0, 2: 8, 3: 8, 4: 13, 5: 20, 6: 18, 8: 3}
    for j, op in want.items():
        assert ores[j].status == 2 and ores[j].opcode_index == op, (j, ores[j].as_tuple())
    assert ores[7].status == 2 and ores[7].opcode_index in (26, 31) and ores[9].opcode_index in (26, 35)
    assert any(x.status == 0 for x in ores) and any(x.status == 2 and x.opcode_index == 26 for x in ores)


@pytest.mark.parametrize("failing", [False, True])
def test_hash_chains(oracle, failing):
    """A byte-message hash of another one's digest runs behind it in the same workgroup (plan.cpp hash chains, kernels_hash.hip): a chain of
    three with the digest bytes interleaved with other inputs and out of order, a second consumer of the same digest (own launch), a
    consumer that also reads a gate output of the head's level and one that reads two digests (no chain), RANGE checks on digest bytes,
    and a chained hash whose output is already assigned (compared)."""
    import acvm_amd
    r = rnd(77)
    n = 48
    ids = list(range(1, n + 1))
    o = n + 1
    d1, d2, d3, d4, d5, d6 = (list(range(o + 32 * k, o + 32 * k + 32)) for k in range(6))
    g = o + 32 * 6
    ops = [BB("SHA256", {"inputs": [FI(w, 8) for w in ids[:20]], "outputs": d1}),                                       # 0
           BB("Keccak256", {"inputs": [FI(w, 8) for w in d1 + ids[20:30]], "outputs": d2}),                             # 1: behind 0
           BB("RANGE", {"input": FI(d2[3], 8)}), BB("RANGE", {"input": FI(d2[4], 5 if failing else 8)}),                  # 2, 3: on digest bytes (3 fails often)
           BB("Blake2s", {"inputs": [FI(w, 8) for w in d2[16:] + ids[30:33] + d2[:16][::-1]], "outputs": d3}),          # 4: behind 1
           BB("SHA256", {"inputs": [FI(w, 8) for w in d1[::-1]], "outputs": d4}),                                       # 5: second consumer of d1
           E([], [(1, g), (P - 1, ids[40])], 0),                                                                        # 6: g = w41, level 1
           BB("Blake2s", {"inputs": [FI(w, 8) for w in d3] + [FI(g, 8)], "outputs": d5}),                               # 7: reads g: no chain
           BB("Keccak256", {"inputs": [FI(w, 8) for w in d4 + d3[:8]], "outputs": d6}),                                 # 8: two digests: no chain
           BB("SHA256", {"inputs": [FI(w, 8) for w in d6[:8] + ids[42:44]], "outputs": [ids[44] if failing else g + 32] + list(range(g + 1, g + 32))})]  # 9: behind 8, first output compared
    circ = Circuit(g + 33, ops)
    rows = [[r.randrange(256) for _ in range(n)] for _ in range(130)]
    rows[3][2] = P - 1          # a field-sized value where a byte is expected: its low byte goes into the message
    rows[4][21] = 1 << 40
    ores, stats = both_paths(oracle, circ, ids, rows)
    assert stats["n_hash_chained"] == 3, stats["n_hash_chained"]
    if failing:
        assert any(x.status == 2 and x.opcode_index == 3 for x in ores) and any(x.status == 2 and x.opcode_index == 9 for x in ores)
    else:
        assert all(x.status == 0 for x in ores)
    with acvm_amd.tuning(hash_chain=0):
        _, stats0 = run_both(oracle, circ, ids, rows)
    assert stats0["n_hash_chained"] == 0


def test_hash_mixed_widths_and_field_inputs(oracle):
    """fetch_nearest_bytes with num_bits != 8: multi-byte little-endian packing, truncation of wide values."""
    r = rnd(7)
    widths = [1, 8, 9, 16, 31, 32, 64, 128, 254, 256, 7, 24]
    n = len(widths)
    ids = list(range(1, n + 1))
    ops = [BB("SHA256", {"inputs": [FI(w, b) for w, b in zip(ids, widths)], "outputs": list(range(20, 52))}),
           BB("HashToField128Security", {"inputs": [FI(w, b) for w, b in zip(ids, widths)], "output": 60}),
           BB("Keccak256", {"inputs": [FI(w, b) for w, b in zip(ids, widths)], "outputs": list(range(70, 102))})]
    circ = Circuit(101, ops)
    rows = [[r.randrange(P) for _ in range(n)] for _ in range(66)]
    both_paths(oracle, circ, ids, rows)


def test_keccak_variable_length(oracle):
    r = rnd(8)
    n = 150
    ids = list(range(1, n + 2))  # last input = var_message_size
    circ = Circuit(n + 40, [BB("Keccak256VariableLength", {"inputs": [FI(w, 8) for w in ids[:n]], "var_message_size": FI(n + 1, 32),
                                                           "outputs": list(range(n + 2, n + 34))})])
    rows = [[r.randrange(256) for _ in range(n)] + [r.randrange(n + 1)] for _ in range(70)]
    rows[0][-1] = 0
    rows[1][-1] = n
    rows[2][-1] = n + 1          # more than the message -> BlackBoxFunctionFailed
    rows[3][-1] = 135
    rows[4][-1] = 136
    rows[5][-1] = (1 << 64) + 5  # `to_u128() as usize` truncates to 5
    rows[6][-1] = 1 << 40
    ores, _ = both_paths(oracle, circ, ids, rows)
    assert ores[2].err == oracle.E_BLACKBOX_FAILED and ores[6].err == oracle.E_BLACKBOX_FAILED and ores[5].status == 0


def test_hash_wrong_output_count(oracle):
    circ = Circuit(40, [BB("SHA256", {"inputs": [FI(1, 8)], "outputs": list(range(2, 33))})])
    ores, _ = both_paths(oracle, circ, [1], [[3], [4]])
    assert ores[0].err == oracle.E_BLACKBOX_FAILED


def test_config3_hash_circuit(oracle):
    from acvm_amd import synth
    circ, ids = synth.hash_circuit()
    B = 300
    values = synth.byte_batch(B, len(ids))
    rows = [[int.from_bytes(values[(j * len(ids) + k) * 32:(j * len(ids) + k + 1) * 32], "big") for k in range(len(ids))] for j in range(B)]
    rows[5][3] = 256  # a RANGE(8) failure
    ores, stats = both_paths(oracle, circ, ids, rows)
    assert ores[5].status == 2 and ores[4].status == 0


def test_config5_mixed_circuit(oracle):
    """The config-5 opcode mix (arithmetic + range / logic + directives + memory + Brillig + hashes + Pedersen) in one circuit:
    level kernels of every class, the exact kernels for the edge-case instances, and every instance through the exact path."""
    from acvm_amd import synth
    circ, ids = synth.mixed_circuit(2500)
    B = 200
    values = synth.witness_batch(B, seed=0xAC1D0005)
    rows = [[int.from_bytes(values[(j * len(ids) + k) * 32:(j * len(ids) + k + 1) * 32], "big") % P for k in range(len(ids))] for j in range(B)]
    ores, stats = run_both(oracle, circ, ids, rows)
    assert sum(1 for j in range(B) if ores[j].status == 0) >= B - 8
    assert stats["n_slow_instances"] <= 8
    run_both(oracle, circ, ids, rows[:48], force_slow=True)
