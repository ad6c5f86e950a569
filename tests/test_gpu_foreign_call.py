"""Brillig foreign-call round trip on the device path: RequiresForeignCall -> get_pending_foreign_call ->
resolve_pending_foreign_call -> solve again (acvm/src/pwg/mod.rs:203-228, acvm/tests/solver.rs:308-426,
acvm_js/test/shared/{foreign_call,complex_foreign_call}.ts as committed fixtures), batch-wise against the CPU oracle."""
import numpy as np
import pytest

import acvm_amd
from acvm_amd.acir import P, Brillig, Circuit, Expression as E
from acvm_amd.synth import values_from_rows

pytestmark = pytest.mark.gpu
M1 = P - 1
W = E.from_witness


def drive(batch, B, respond):
    """Solve, answering every pending call with respond(instance, function, inputs) until nothing waits. Returns rounds."""
    rounds = 0
    while True:
        batch.solve()
        res = batch.results()
        waiting = [j for j in range(B) if res[j].status == acvm_amd.STATUS_REQUIRES_FOREIGN_CALL]
        if not waiting:
            return rounds, res
        rounds += 1
        assert rounds < 50
        for j in waiting:
            fn, inputs = batch.get_pending_foreign_call(j)
            batch.resolve_pending_foreign_call(j, respond(j, fn, inputs))


@pytest.mark.parametrize("name", ["foreign_call", "complex_foreign_call"])
def test_acvm_js_fixture(golden, name):
    fx = golden["acvm_js"][name]
    iw = {int(k): int(v, 16) for k, v in fx["initialWitnessMap"].items()}
    ids = sorted(iw)
    batch = acvm_amd.Batch(acvm_amd.Circuit(bytes(fx["bytecode"])), 1, ids)
    batch.set_initial_witness(values_from_rows([[iw[i] for i in ids]]))

    def respond(j, fn, inputs):
        assert fn == fx["oracleCallName"]
        assert inputs == [[int(x, 16) for x in grp] for grp in fx["oracleCallInputs"]]
        return [int(x, 16) if isinstance(x, str) else [int(y, 16) for y in x] for x in fx["oracleResponse"]]

    rounds, res = drive(batch, 1, respond)
    assert rounds >= 1 and res[0].status == acvm_amd.STATUS_SOLVED
    asg, vals = batch.witness_map()
    got = {w: int.from_bytes(vals[0, w].tobytes(), "big") for w in range(asg.shape[1]) if asg[0, w]}
    assert got == {int(k): int(v, 16) for k, v in fx["expectedWitnessMap"].items()}


def _oracle_run(oracle, circ_bytes, ids, row, respond, j):
    a = oracle.ACVM(oracle.Circuit(circ_bytes), dict(zip(ids, row)))
    st = a.solve()
    while st == oracle.ST_REQUIRES_FOREIGN_CALL:
        fn, inputs = a.get_pending_foreign_call()
        a.resolve_pending_foreign_call(respond(j, fn, inputs))
        st = a.solve()
    return a


def test_oracle_dependent_execution_batch(oracle):
    """solver.rs:308-426 shape: two `invert` calls inside one Brillig opcode, B instances, arithmetic before and after; some
    instances fail before the Brillig opcode, some never wait (predicate 0), the others wait twice."""
    br = Brillig(inputs=[W(1), E(), W(2)], outputs=[5, 6, 7, 8],
                 bytecode=[("ForeignCall", "invert", [("Register", 1)], [("Register", 0)]),
                           ("ForeignCall", "invert", [("Register", 3)], [("Register", 2)])], predicate=W(3))
    circ = Circuit(10, [E([(1, 1, 2)], [(M1, 4)], 0),      # w4 = w1 * w2
                        E([], [(1, 4), (M1, 9)], 1),        # w9 = w4 + 1
                        br,
                        E([(1, 1, 6)], [(M1, 10)], 0)])     # w10 = w1 * w6 (= 1 when the call ran)
    ids = [1, 2, 3]
    rows = [[3 + j, 7 * j + 1, 1 if j % 4 else 0] for j in range(70)]
    data = circ.to_bytes()

    def respond(j, fn, inputs):
        assert fn == "invert" and len(inputs) == 1 and len(inputs[0]) == 1
        return [pow(inputs[0][0], P - 2, P)]

    batch = acvm_amd.Batch(acvm_amd.Circuit(data), len(rows), ids)
    batch.set_initial_witness(values_from_rows(rows))
    rounds, res = drive(batch, len(rows), respond)
    assert rounds == 2
    asg, vals = batch.witness_map()
    for j, row in enumerate(rows):
        a = _oracle_run(oracle, data, ids, row, respond, j)
        r = a.result()
        assert res[j].as_tuple() == r.as_tuple(), j
        wm = a.witness_map()
        got = {w: int.from_bytes(vals[j, w].tobytes(), "big") for w in range(asg.shape[1]) if asg[j, w]}
        assert got == wm, j
    with pytest.raises(acvm_amd.AcvmError):
        batch.resolve_pending_foreign_call(0, [1])  # nothing is waiting any more: the reference panics


def test_array_inputs_outputs_and_two_brillig_opcodes(oracle):
    """HeapArray / HeapVector inputs and destinations; a second Brillig opcode with its own call must not see the first
    opcode's results."""
    b1 = Brillig(inputs=[[W(1), W(2), W(3)], W(4)], outputs=[[10, 11, 12], 13],
                 bytecode=[("Const", 2, 3), ("Const", 3, 20),
                           ("ForeignCall", "sort", [("HeapArray", 3, 3), ("Register", 4)], [("HeapVector", 0, 2), ("Register", 1), ("HeapArray", 0, 2)]),
                           ("Mov", 0, 3), ("Mov", 1, 4), ("Stop",)])
    b2 = Brillig(inputs=[W(13)], outputs=[14, [15, 16]],
                 bytecode=[("Const", 5, 30), ("ForeignCall", "twice", [("Register", 1), ("HeapVector", 5, 6)], [("Register", 0)]),
                           ("Mov", 0, 1), ("Mov", 1, 5), ("Stop",)])
    circ = Circuit(16, [b1, b2])
    ids = [1, 2, 3, 4]
    rows = [[(5 * j + 3) % 17, (3 * j + 1) % 17, (7 * j) % 17, j] for j in range(66)]
    data = circ.to_bytes()

    def respond(j, fn, inputs):
        if fn == "sort":
            assert [len(x) for x in inputs] == [3, 1, 2]
            return [sorted(inputs[0]), (sum(inputs[0]) + inputs[1][0]) % P]
        assert fn == "twice" and len(inputs) == 1
        return [2 * inputs[0][0] % P, [inputs[0][0], 7]]

    batch = acvm_amd.Batch(acvm_amd.Circuit(data), len(rows), ids)
    batch.set_initial_witness(values_from_rows(rows))
    rounds, res = drive(batch, len(rows), respond)
    assert rounds == 2
    asg, vals = batch.witness_map()
    for j, row in enumerate(rows):
        a = _oracle_run(oracle, data, ids, row, respond, j)
        assert res[j].as_tuple() == a.result().as_tuple(), j
        got = {w: int.from_bytes(vals[j, w].tobytes(), "big") for w in range(asg.shape[1]) if asg[j, w]}
        assert got == a.witness_map(), j


def test_wrong_result_shapes(oracle):
    """lib.rs:262-270: a result with the wrong number of values / wrong array size fails the VM (or finishes it when the
    call is the last instruction)."""
    last = Brillig(inputs=[W(1)], outputs=[2], bytecode=[("ForeignCall", "f", [("Register", 0), ("Register", 1)], [("Register", 0)])])
    mid = Brillig(inputs=[W(1)], outputs=[3], bytecode=[("ForeignCall", "g", [("HeapArray", 0, 2)], [("Register", 0)]), ("Stop",)])
    for circ, answer in [(Circuit(3, [last]), [5]), (Circuit(3, [mid]), [[1, 2, 3]]), (Circuit(3, [mid]), [4])]:
        data = circ.to_bytes()
        rows = [[9], [10]]
        respond = lambda j, fn, inputs: answer  # noqa: E731
        batch = acvm_amd.Batch(acvm_amd.Circuit(data), 2, [1])
        batch.set_initial_witness(values_from_rows(rows))
        _, res = drive(batch, 2, respond)
        for j, row in enumerate(rows):
            a = _oracle_run(oracle, data, [1], row, respond, j)
            r = a.result()
            assert res[j].as_tuple() == r.as_tuple(), (j, res[j].as_tuple(), r.as_tuple())
            assert res[j].message == r.message
            assert list(res[j].call_stack[:res[j].n_call_stack]) == list(r.call_stack[:r.n_call_stack])
