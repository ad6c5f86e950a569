"""Config 5's outputs without the full maps: the per-instance digest folded into the solve (ACVM_BATCH_FOLD_DIGEST) and
witness-slot liveness reuse (ACVM_BATCH_REUSE_SLOTS) against the plain batch and, through the digest (Blake2s over the whole map,
recomputed with hashlib over the oracle's map), against the reference's finalize() map (acvm/src/pwg/mod.rs:176-181)."""
import numpy as np
import pytest

import acvm_amd
from acvm_amd import synth

pytestmark = pytest.mark.gpu


def plain(data, ids, values, B):
    gc = acvm_amd.Circuit(data)
    b = acvm_amd.Batch(gc, B, ids)
    b.set_initial_witness(values)
    b.solve()
    return gc, b


@pytest.mark.parametrize("gates,B", [(400, 96), (3000, 700)])
def test_folded_digest_equals_digest_after_the_solve(oracle, gates, B):
    circ, ids = synth.mixed_circuit(gates, seed=0xAC1D0F01)
    data = circ.to_bytes()
    values = synth.witness_batch(B, seed=0xAC1D0F01)  # instances 0..7 are the edge cases: some leave the generic path
    gc, b0 = plain(data, ids, values, B)
    want = b0.digest()
    b1 = acvm_amd.Batch(gc, B, ids, fold_digest=True)
    b1.set_initial_witness(values)
    b1.solve()
    assert [r.as_tuple() for r in b1.results()] == [r.as_tuple() for r in b0.results()]
    assert b1.stats()["n_slow_instances"] == b0.stats()["n_slow_instances"] > 0
    assert np.array_equal(b1.digest(), want)
    assert np.array_equal(b1.digest(5, 40), want[5:45])
    # the full maps are still there in this mode
    a0, v0 = b0.witness_map(0, 16)
    a1, v1 = b1.witness_map(0, 16)
    assert np.array_equal(a0, a1) and np.array_equal(v0, v1)
    # and the digest is the oracle's map's
    row = len(ids) * 32
    picks = [0, 3, 9, B - 1]
    ores, oasg, ovals = oracle.solve_batch(oracle.Circuit(data), ids, b"".join(values[j * row:(j + 1) * row] for j in picks), len(picks))
    for i, j in enumerate(picks):
        assert bytes(want[j]) == oracle.witness_map_digest(oasg[i], ovals[i]), j
    # a second set of inputs through the same handle
    values2 = synth.witness_batch(B, seed=0xAC1D0F02, edge_cases=False)
    b1.set_initial_witness(values2)
    b1.solve()
    b0.set_initial_witness(values2)
    b0.solve()
    assert np.array_equal(b1.digest(), b0.digest())


@pytest.mark.parametrize("gates,B", [(400, 96), (3000, 700), (20000, 300)])
def test_slot_reuse_against_the_plain_batch(oracle, gates, B):
    """results, return witnesses and digests of a batch that recycles witness rows = those of the plain batch (whose maps the other
    tests pin to the oracle); the edge-case instances are re-solved from their initial witnesses in the exact path's own table"""
    circ, ids = synth.mixed_circuit(gates, seed=0xAC1D0F03)
    data = circ.to_bytes()
    values = synth.witness_batch(B, seed=0xAC1D0F03)
    gc, b0 = plain(data, ids, values, B)
    ret = gc.witness_set("return_values")
    # keep the return value and a witness from the middle of the circuit
    keep = ret + [gc.num_witnesses // 2]
    b1 = acvm_amd.Batch(gc, B, ids, reuse_slots=True, keep=keep)
    st = b1.stats()
    assert st["n_table_rows"] < st["n_witnesses"] and st["n_digest_segments"] > 0
    b1.set_initial_witness(values)
    b1.solve()
    res0, res1 = b0.results(), b1.results()
    assert [r.as_tuple() for r in res1] == [r.as_tuple() for r in res0]
    assert [r.message for r in res1] == [r.message for r in res0]
    assert b1.stats()["n_slow_instances"] == b0.stats()["n_slow_instances"] > 0
    assert np.array_equal(b1.digest(), b0.digest())
    for w in keep + ids[:2]:
        v0, a0 = b0.witness(w)
        v1, a1 = b1.witness(w)
        assert np.array_equal(a0, a1) and np.array_equal(v0, v1), w
    solved = [j for j in range(B) if res0[j].status == 0]
    j0 = solved[0]
    assert np.array_equal(b1.extract(keep, j0, 1), b0.extract(keep, j0, 1))
    with pytest.raises(acvm_amd.AcvmError, match="not kept"):
        b1.witness(ids[-1] + 3)
    with pytest.raises(acvm_amd.AcvmError, match="recycles"):
        b1.witness_map(0, 1)
    # the handle is reusable: other inputs, no edge cases
    values2 = synth.witness_batch(B, seed=0xAC1D0F04, edge_cases=False)
    for b in (b0, b1):
        b.set_initial_witness(values2)
        b.solve()
    assert np.array_equal(b1.digest(), b0.digest())
    assert np.array_equal(b1.extract(ret, 0, B), b0.extract(ret, 0, B))
    # and the oracle agrees on the digests of a sample
    row = len(ids) * 32
    picks = [1, B // 2]
    ores, oasg, ovals = oracle.solve_batch(oracle.Circuit(data), ids, b"".join(values2[j * row:(j + 1) * row] for j in picks), len(picks))
    dig = b1.digest()
    for i, j in enumerate(picks):
        assert bytes(dig[j]) == oracle.witness_map_digest(oasg[i], ovals[i]), j


def test_slot_reuse_refusals():
    from acvm_amd.acir import P, Brillig, Circuit, Expression as E
    circ = Circuit(3, [Brillig(inputs=[E.from_witness(1)], outputs=[2], bytecode=[("ForeignCall", "f", [("Register", 0)], [("Register", 0)]), ("Stop",)])])
    with pytest.raises(acvm_amd.AcvmError, match="slot reuse"):
        acvm_amd.Batch(acvm_amd.Circuit(circ.to_bytes()), 4, [1], reuse_slots=True)


def test_kept_witness_that_nothing_produces_reads_as_unassigned():
    """slot reuse: a keep id below n_witnesses that no opcode assigns has no row of the table; reading it back must answer "unassigned, zero"
    like the plain batch does (round 2 read terabytes past the table)"""
    from acvm_amd.acir import P, Circuit, Expression as E
    from acvm_amd.synth import values_from_rows
    circ = Circuit(9, [E([(1, 1, 2)], [(P - 1, 3)], 0)])  # witnesses 4..9 exist in the numbering but nothing assigns them
    gc = acvm_amd.Circuit(circ.to_bytes())
    for reuse in (False, True):
        b = acvm_amd.Batch(gc, 70, [1, 2], reuse_slots=reuse, keep=[3, 7])
        b.set_initial_witness(values_from_rows([[j + 1, j + 2] for j in range(70)]))
        assert b.solve() == 0
        vals, asg = b.witness(7)
        assert not asg.any() and not vals.any()
        vals, asg = b.witness(3)
        assert asg.all() and int.from_bytes(vals[5].tobytes(), "big") == 6 * 7
        with pytest.raises(acvm_amd.AcvmError, match="Witness not found"):
            b.extract([3, 7])
