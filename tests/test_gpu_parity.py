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
