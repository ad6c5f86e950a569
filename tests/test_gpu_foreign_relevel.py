"""Foreign calls answered for the whole batch re-enter the LEVEL schedule (the resolved results live in a per-opcode, per-instance
store the Brillig level kernel reads): same final witness maps as the exact in-order resume and as the CPU oracle driven through
ACVM::resolve_pending_foreign_call (acvm/src/pwg/mod.rs:203-228, brillig_vm/src/lib.rs:190-274), two sequential oracle calls per
instance, and the opcodes behind the calls solved by the level kernels (no instance left on the exact path)."""
import os

import numpy as np
import pytest

import acvm_amd
from acvm_amd.acir import P, Brillig, Circuit, Expression as E
from acvm_amd.synth import values_from_rows

pytestmark = pytest.mark.gpu
W = E.from_witness


def circuit(n_tail=40):
    """w3 = oracle 'invert'(w1) ; w4 = w3 * w1 (must be 1) ; w5 = oracle 'double'(w2 + w4) [2 results] ; then a tail of gates"""
    ops = [Brillig(inputs=[W(1)], outputs=[3], bytecode=[("ForeignCall", "invert", [("Register", 0)], [("Register", 0)]), ("Stop",)]),
           E([(1, 3, 1)], [(P - 1, 4)], 0),
           Brillig(inputs=[E([], [(1, 2), (1, 4)], 0)], outputs=[5, 6],
                   bytecode=[("ForeignCall", "double", [("Register", 0), ("Register", 1)], [("Register", 0)]), ("Stop",)])]
    nw = 6
    for i in range(n_tail):  # w_{k} = w_{k-1} * w_{k-2} + w5
        ops.append(E([(1, nw, nw - 1)], [(1, 5), (P - 1, nw + 1)], 0))
        nw += 1
    return Circuit(nw, ops), [1, 2]


def respond(fn, inputs):
    if fn == "invert":
        return [pow(inputs[0][0], P - 2, P)]
    assert fn == "double"
    return [inputs[0][0] * 2 % P, (inputs[0][0] * 2 + 1) % P]


def drive(batch, B):
    rounds = 0
    while True:
        batch.solve()
        res = batch.results()
        waiting = [j for j in range(B) if res[j].status == acvm_amd.STATUS_REQUIRES_FOREIGN_CALL]
        if not waiting:
            return rounds, res
        rounds += 1
        assert rounds < 10
        for j in waiting:
            fn, inputs = batch.get_pending_foreign_call(j)
            batch.resolve_pending_foreign_call(j, respond(fn, inputs))


@pytest.mark.parametrize("mode", ["relevel", "exact"])
def test_relevel_equals_exact_resume_and_oracle(oracle, mode):
    circ, ids = circuit()
    data = circ.to_bytes()
    B = 96
    rows = [[3 + j, 1000 + 7 * j] for j in range(B)]
    rows[5][0] = 0  # invert(0) = 0: w4 = 0, still solvable
    with acvm_amd.tuning(fc_relevel=1 if mode == "relevel" else 0):
        batch = acvm_amd.Batch(acvm_amd.Circuit(data), B, ids)
        batch.set_initial_witness(values_from_rows(rows))
        rounds, res = drive(batch, B)
    assert rounds == 2
    st = batch.stats()
    if mode == "relevel":
        assert st["n_slow_instances"] == 0  # everything behind the calls ran on the level kernels
    asg, vals = batch.witness_map()
    oc = oracle.Circuit(data)
    for j in range(B):
        a = oracle.ACVM(oc, dict(zip(ids, rows[j])))
        s = a.solve()
        while s == oracle.ST_REQUIRES_FOREIGN_CALL:
            fn, inputs = a.get_pending_foreign_call()
            a.resolve_pending_foreign_call(respond(fn, inputs))
            s = a.solve()
        assert res[j].as_tuple() == a.result().as_tuple(), j
        got = {w: int.from_bytes(vals[j, w].tobytes(), "big") for w in range(asg.shape[1]) if asg[j, w]}
        assert got == a.witness_map(), j
    # a fresh set of inputs forgets the answers: every instance waits again
    batch.set_initial_witness(values_from_rows(rows))
    batch.solve()
    assert all(r.status == acvm_amd.STATUS_REQUIRES_FOREIGN_CALL for r in batch.results())


def test_partially_answered_batch():
    """only half of the instances are answered before the next solve: the others keep waiting at the same call with the same inputs"""
    circ, ids = circuit(4)
    B = 32
    rows = [[3 + j, 50 + j] for j in range(B)]
    batch = acvm_amd.Batch(acvm_amd.Circuit(circ.to_bytes()), B, ids)
    batch.set_initial_witness(values_from_rows(rows))
    batch.solve()
    for j in range(0, B, 2):
        fn, inputs = batch.get_pending_foreign_call(j)
        batch.resolve_pending_foreign_call(j, respond(fn, inputs))
    batch.solve()
    res = batch.results()
    for j in range(B):
        fn, inputs = batch.get_pending_foreign_call(j)
        assert fn == ("double" if j % 2 == 0 else "invert"), j
        assert res[j].opcode_index == (2 if j % 2 == 0 else 0)
        if j % 2:
            assert inputs == [[3 + j]]
