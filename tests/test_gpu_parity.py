"""GPU parity tests proper: the HIP path (through the C ABI of include/acvm_amd.h) against the CPU oracle on
identical seeded inputs. Bit-exact bar: status, error kind, failing opcode index, assigned set and every
32-byte witness value of every instance."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run_both(oracle, circ, ids, values, B, force_slow=False):
    import acvm_amd
    data = circ.to_bytes()
    oc = oracle.Circuit(data)
    ores, oasg, ovals = oracle.solve_batch(oc, ids, values, B)
    gc = acvm_amd.Circuit(data)
    batch = acvm_amd.Batch(gc, B, ids)
    batch.set_force_slow_path(force_slow)
    batch.set_initial_witness(values)
    batch.solve()
    gres = batch.results()
    gasg, gvals = batch.witness_map()
    stats = batch.stats()
    batch.free()
    return (ores, oasg, ovals), (gres, gasg, gvals), stats


def _assert_parity(o, g, B):
    ores, oasg, ovals = o
    gres, gasg, gvals = g
    for j in range(B):
        assert gres[j].as_tuple() == ores[j].as_tuple(), f"instance {j}: gpu {gres[j].as_tuple()} oracle {ores[j].as_tuple()}"
    nw = min(oasg.shape[1], gasg.shape[1])
    assert np.array_equal(oasg[:, :nw], gasg[:, :nw]), "assigned sets differ"
    assert np.array_equal(ovals[:, :nw], gvals[:, :nw]), "witness values differ"


@pytest.mark.parametrize("n_gates,B", [(64, 70), (1000, 256), (3000, 96)])
def test_arithmetic_circuit_parity(oracle, n_gates, B):
    from acvm_amd import synth
    circ, ids = synth.arithmetic_circuit(n_gates, seed=0xAC1D0002 + n_gates)
    values = synth.witness_batch(B, seed=0xAC1D0002 + n_gates)
    o, g, stats = _run_both(oracle, circ, ids, values, B)
    _assert_parity(o, g, B)
    solved = sum(1 for j in range(B) if g[0][j].status == 0)
    assert solved >= B - 8  # only the edge-case instances may fail / leave the generic path
    assert stats["n_slow_instances"] <= 8


@pytest.mark.parametrize("force_slow", [False, True])
def test_wide_gate_parity(oracle, force_slow):
    """Up to 12 mul + 12 linear terms per gate: several column carries of the device's sum-of-products accumulator, unit
    coefficients beyond the side-sum budget, assert-zero gates."""
    from acvm_amd import synth
    circ, ids = synth.wide_gate_circuit(240)
    values = synth.witness_batch(96, seed=0xAC1D0A11)
    o, g, stats = _run_both(oracle, circ, ids, values, 96, force_slow=force_slow)
    _assert_parity(o, g, 96)
    assert sum(1 for j in range(96) if g[0][j].status == 0) >= 88


def test_chain_circuit_parity(oracle):
    from acvm_amd import synth
    circ, ids = synth.arithmetic_circuit(300, seed=0xAC1D0077, chain=True)
    values = synth.witness_batch(64, seed=0xAC1D0077)
    o, g, stats = _run_both(oracle, circ, ids, values, 64)
    _assert_parity(o, g, 64)
    assert stats["n_levels"] >= 300


def test_exact_inorder_kernel_parity(oracle):
    """Every instance through the exact in-order kernel (the path taken by failing / non-generic instances)."""
    from acvm_amd import synth
    circ, ids = synth.arithmetic_circuit(400, seed=0xAC1D0123)
    values = synth.witness_batch(80, seed=0xAC1D0123)
    o, g, stats = _run_both(oracle, circ, ids, values, 80, force_slow=True)
    _assert_parity(o, g, 80)
    assert stats["n_slow_instances"] == 80


def test_reference_addition_fixture(oracle, golden):
    import acvm_amd
    fx = golden["acvm_js"]["addition"]
    gc = acvm_amd.Circuit(bytes(fx["bytecode"]))
    iw = {int(k): int(v, 16) for k, v in fx["initialWitnessMap"].items()}
    ids = sorted(iw)
    batch = acvm_amd.Batch(gc, 1, ids)
    batch.set_initial_witness(b"".join(iw[i].to_bytes(32, "big") for i in ids))
    assert batch.solve() == 0
    vals, asg = batch.witness(fx["resultWitness"])
    assert asg[0] == 1 and int.from_bytes(vals[0].tobytes(), "big") == int(fx["expectedResult"], 16)


def test_unsatisfied_constraint_reports_opcode(oracle):
    """acvm/tests/solver.rs:490-525: x == y with 1 != 2 fails with UnsatisfiedConstrain at opcode 0."""
    import acvm_amd
    from acvm_amd.acir import Circuit, Expression, P
    circ = Circuit(2, [Expression([], [(1, 1), (P - 1, 2)], 0)])
    o, g, _ = _run_both(oracle, circ, [1, 2], (1).to_bytes(32, "big") + (2).to_bytes(32, "big"), 1)
    _assert_parity(o, g, 1)
    assert g[0][0].as_tuple() == (acvm_amd.STATUS_FAILURE, acvm_amd.ERR_UNSATISFIED, 0, 0, 0)


@pytest.mark.parametrize("pattern", ["all", "every_third", "one_per_wave", "none"])
def test_flagged_instances_are_counted_by_the_kernels_that_flag(oracle, pattern):
    """Since round 6 no kernel counts the flagged instances behind a solve: the kernels that flag count (the first flag of an instance, one pair of
    atomics per wave). A batch in which every instance fails at the same opcode, every third one, one lane of every wave, none: the exact path gets
    exactly those instances (statistics), results and maps equal the oracle's; a second solve of the same handle with other inputs counts afresh, and
    the pipelined form (acvm_batch_solve_then_import) holds its import back exactly when something was flagged."""
    import acvm_amd
    from acvm_amd.acir import Circuit, Expression, P
    # w3 = w1 * w2 (a gate), then assert w3 == w4 (flags where it does not hold), then RANGE(w3, 64) (flags large products)
    from acvm_amd.acir import BlackBoxFuncCall as BB, FunctionInput as FI
    circ = Circuit(5, [Expression([(1, 1, 2)], [(P - 1, 3)], 0), Expression([], [(1, 3), (P - 1, 4)], 0), BB("RANGE", {"input": FI(3, 64)}),
                       Expression([], [(1, 3), (1, 1), (P - 1, 5)], 0)])
    B = 5000  # 78 waves and a partial one
    fails = {"all": lambda j: True, "every_third": lambda j: j % 3 == 0, "one_per_wave": lambda j: j % 64 == 17, "none": lambda j: False}[pattern]
    rows = []
    for j in range(B):
        a, b = j + 2, 3 * j + 1
        rows.append([a, b, a * b + (1 if fails(j) else 0)])
    values = b"".join(x.to_bytes(32, "big") for r in rows for x in r)
    ids = [1, 2, 4]
    o, g, st = _run_both(oracle, circ, ids, values, B)
    _assert_parity(o, g, B)
    n_fail = sum(1 for j in range(B) if fails(j))
    assert st["n_slow_instances"] == n_fail and sum(1 for r in g[0] if r.status != 0) == n_fail
    # the same handle again, the other way round, then pipelined against a resident buffer
    gc = acvm_amd.Circuit(circ.to_bytes())
    batch = acvm_amd.Batch(gc, B, ids)
    clean = b"".join(x.to_bytes(32, "big") for j in range(B) for x in (j + 2, 3 * j + 1, (j + 2) * (3 * j + 1)))
    buf_bad, buf_clean = acvm_amd.DeviceBuffer(values), acvm_amd.DeviceBuffer(clean)
    for first, second, want in ((buf_bad, buf_clean, n_fail), (buf_clean, buf_bad, 0), (buf_clean, buf_clean, 0), (buf_bad, buf_bad, n_fail)):
        batch.set_initial_witness_device(first.ptr)
        assert batch.solve(then_import=second.ptr) == want and batch.stats()["n_slow_instances"] == want
        batch.set_initial_witness_device(second.ptr)  # (free when the import ran behind the solve; performed now when it was held back)
        want2 = n_fail if second is buf_bad else 0
        assert batch.solve() == want2 and batch.stats()["n_slow_instances"] == want2
    batch.free()
    buf_bad.free()
    buf_clean.free()


def test_empty_and_ragged(oracle):
    import acvm_amd
    from acvm_amd import synth
    from acvm_amd.acir import Circuit
    # empty opcode list: Solved at construction (pwg/mod.rs:147)
    gc = acvm_amd.Circuit(Circuit(1, []).to_bytes())
    b = acvm_amd.Batch(gc, 3, [1])
    b.set_initial_witness(b"\x00" * 96)
    assert b.solve() == 0
    assert all(r.status == 0 for r in b.results())
    # batch sizes that are not multiples of the wavefront / block
    circ, ids = synth.arithmetic_circuit(50, seed=5)
    for B in (1, 63, 65, 257):
        values = synth.witness_batch(B, seed=5, edge_cases=False)
        o, g, _ = _run_both(oracle, circ, ids, values, B)
        _assert_parity(o, g, B)


def test_device_field_selftest():
    """hand-scheduled gfx950 field routines against the portable ones on 2^18 random operand pairs"""
    import acvm_amd
    assert acvm_amd.selftest(1 << 18, 7) == 0


def test_config1_fixture():
    """BASELINE config 1 on the device, as a batch of one and inside a batch of 70: against tests/golden/config1.json
    (independent Python big-integer solve, tests/golden/make_config1_fixture.py)."""
    import hashlib
    import json
    import os
    import acvm_amd
    from acvm_amd import synth
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config1.json")))
    circ, ids = synth.arithmetic_circuit(fx["gates"], seed=fx["seed"])
    row = bytes.fromhex(fx["inputs_be32_hex"])
    nw = fx["n_witnesses"]
    for B, at in ((1, 0), (70, 37)):
        values = bytearray(synth.witness_batch(B, seed=fx["seed"]))
        values[at * len(row):(at + 1) * len(row)] = row
        batch = acvm_amd.Batch(acvm_amd.Circuit(circ.to_bytes()), B, ids)
        batch.set_initial_witness(bytes(values))
        batch.solve()
        assert batch.results()[at].status == 0
        asg, vals = batch.witness_map(at, 1)
        assert asg[0, 1:nw + 1].all()
        assert hashlib.sha256(vals[0, 1:nw + 1].tobytes()).hexdigest() == fx["sha256_of_witnesses_1_to_n"]
        for w, v in fx["witnesses"].items():
            assert vals[0, int(w)].tobytes().hex() == v
        batch.free()


def test_random_circuit_shapes(oracle):
    """Seeded sweep over circuit shapes the planner treats differently: gate mixes with up to 40 % inversion gates (long
    chains of early inversion batches, inverse-row reuse), chains (every level one gate: pairs, batches nobody waits for),
    wide gates, tiny and ragged batch sizes. Each against the oracle, bit for bit."""
    import random
    from acvm_amd import synth
    r = random.Random(0xAC1D)
    for case in range(36):
        n_gates = r.choice([1, 2, 7, 60, 150, 400])
        dyn = r.choice([0, 5, 20, 40])
        rest = 100 - dyn
        a = r.randrange(0, rest + 1)
        b = r.randrange(0, rest - a + 1)
        mix = (a, b, rest - a - b, dyn)
        chain = r.random() < 0.3
        B = r.choice([1, 2, 63, 64, 65, 130])
        seed = 0xAC1D1000 + case
        if case % 6 == 5:
            circ, ids = synth.wide_gate_circuit(n_gates, seed=seed, max_terms=r.choice([3, 8, 15]))
        else:
            circ, ids = synth.arithmetic_circuit(n_gates, seed=seed, chain=chain, mix=mix)
        values = synth.witness_batch(B, seed=seed, edge_cases=(case % 2 == 0))
        o, g, stats = _run_both(oracle, circ, ids, values, B)
        _assert_parity(o, g, B)


def test_unknown_in_mul_and_linear_term(oracle):
    """The unknown both in a mul term (known partner) and in a linear term. `ArithmeticSolver::evaluate` (arithmetic.rs:212-239) turns the mul
    term into a second LINEAR term on the same witness, `solve_fan_in_term` counts terms, not witnesses, so the reference answers
    ExpressionHasTooManyUnknowns (the `w1 == w2` arm of `solve`, :43-62, is unreachable behind `evaluate`) -- unless the partner makes the
    folded coefficient zero, in which case the term is dropped and the linear term alone solves the witness. The planner stops the level
    schedule at such an opcode (every generic instance fails there); the exact kernels must give the reference's answer per instance."""
    from acvm_amd.acir import Circuit, Expression as E, P
    import acvm_amd
    ops = [
        E([(5, 1, 2)], [(P - 1, 4)], 11),                          # w4 = 5 w1 w2 + 11
        E([(7, 4, 5)], [(3, 5), (2, 3), (9, 1)], 13),              # w5 (7 w4) + 3 w5 + 2 w3 + 9 w1 + 13 = 0
        E([(1, 5, 5)], [(P - 1, 6)], 3),                           # w6 = w5^2 + 3
    ]
    circ = Circuit(6, ops)
    ids = [1, 2, 3]
    B = 40
    rng = np.random.default_rng(11)
    rows = [[int.from_bytes(rng.bytes(31), "big") for _ in range(3)] for _ in range(B)]
    rows[2] = [1, ((-11) * pow(5, -1, P)) % P, 7]                   # w4 == 0: the mul term vanishes, 3 w5 + ... solves w5
    rows[3] = [0, 5, 0]                                            # w4 == 11
    values = b"".join(b"".join(v.to_bytes(32, "big") for v in r) for r in rows)
    for force_slow in (False, True):
        o, g, stats = _run_both(oracle, circ, ids, values, B, force_slow=force_slow)
        _assert_parity(o, g, B)
        assert g[0][2].status == acvm_amd.STATUS_SOLVED
        for j in (0, 1, 3, 4, B - 1):
            assert g[0][j].as_tuple()[:3] == (acvm_amd.STATUS_FAILURE, acvm_amd.ERR_TOO_MANY_UNKNOWNS, 1), g[0][j].as_tuple()


def test_projective_witnesses_hand_over(oracle):
    """plan.cpp keeps arithmetic-only witnesses as scale x value. Checked here where the scaled columns meet everything else:
    a constraint on scaled witnesses that fails for some instances in mid-circuit (those columns are unscaled for the exact
    kernels, the others on export), an inversion gate whose denominator and numerator are scaled, and a black-box opcode
    reading a witness between scaled ones (pinned to scale 1)."""
    from acvm_amd.acir import BlackBoxFuncCall as BB, Circuit, Expression as E, FunctionInput as FI, P
    import acvm_amd
    ops = [
        E([(5, 1, 2)], [(P - 1, 3)], 11),                  # w3 = 5 w1 w2 + 11                (scaled)
        E([(7, 3, 1)], [(3, 2), (P - 9, 4)], 0),           # 9 w4 = 7 w3 w1 + 3 w2            (scaled, reads a scaled witness)
        E([(2, 4, 5)], [(13, 3)], 1),                      # 2 w4 w5 + 13 w3 + 1 = 0          (inversion gate, scaled denominator)
        E([], [(1, 5), (P - 1, 6)], 0),                    # w6 = w5                          (read by RANGE below: pinned)
        BB("RANGE", {"input": FI(6, 254)}),
        E([(3, 5, 6)], [(P - 1, 7)], 0),                   # w7 = 3 w5 w6
        E([(1, 7, 7)], [(P - 4, 8)], 5),                   # 4 w8 = w7^2 + 5
        E([], [(6, 8), (P - 6, 9)], 0),                    # constraint on scaled witnesses: fails unless w9 == w8
        E([(11, 8, 3)], [(P - 1, 10)], 0),                 # after the failing opcode
    ]
    circ = Circuit(10, ops)
    ids = [1, 2, 9]
    B = 96
    # solve once on the oracle with a dummy w9 to learn w8, then feed w9 = w8 to every instance but each 5th
    rng = np.random.default_rng(7)
    rows = [[int.from_bytes(rng.bytes(31), "big") for _ in range(2)] for _ in range(B)]
    rows[3] = [0, 5]      # w3 = 11, w4 = 15/9, fine; zero products
    rows[4] = [1, 0]
    rows[6] = [1, (-77 * pow(38, -1, P)) % P]   # makes w4 = 0: the inversion gate's (scaled) denominator vanishes at opcode 2
    probe = b"".join(b"".join(v.to_bytes(32, "big") for v in r + [0]) for r in rows)
    ores, oasg, ovals = oracle.solve_batch(oracle.Circuit(circ.to_bytes()), ids, probe, B)
    w8 = [int.from_bytes(bytes(ovals[j, 8]), "big") for j in range(B)]
    values = b"".join(b"".join(v.to_bytes(32, "big") for v in rows[j] + [w8[j] if j % 5 else (w8[j] + 1) % P]) for j in range(B))
    for force_slow in (False, True):
        o, g, stats = _run_both(oracle, circ, ids, values, B, force_slow=force_slow)
        _assert_parity(o, g, B)
        failed = [j for j in range(B) if g[0][j].status != 0]
        assert set(range(0, B, 5)) <= set(failed)
        for j in range(0, B, 5):
            assert g[0][j].as_tuple()[:3] == (acvm_amd.STATUS_FAILURE, acvm_amd.ERR_UNSATISFIED, 7)
        assert g[0][6].as_tuple()[:3] == (acvm_amd.STATUS_FAILURE, acvm_amd.ERR_UNSATISFIED, 2)
        if not force_slow:
            assert stats["n_scaled_witnesses"] >= 4
