"""The BlackBoxFunctionSolver trait as a caller-supplied vtable (acvm_bb_solver_t): the reference's two fakes
(DummyBlackBoxSolver brillig_vm/src/lib.rs:392-420, a failing backend) and a real backend delegated to host code, on the
level path and on the exact path, against the CPU oracle run with the same backend."""
import ctypes as C

import numpy as np
import pytest

import acvm_amd
from acvm_amd.acir import P, BlackBoxFuncCall as BB, Circuit, Expression as E, FunctionInput as FI
from acvm_amd.synth import grumpkin_circuit, grumpkin_rows, values_from_rows

pytestmark = pytest.mark.gpu


def solve(circ, ids, rows, solver, force_slow=False):
    batch = acvm_amd.Batch(acvm_amd.Circuit(circ.to_bytes()), len(rows), ids, solver=solver)
    batch.set_force_slow_path(force_slow)
    batch.set_initial_witness(values_from_rows(rows))
    batch.solve()
    res = batch.results()
    asg, vals = batch.witness_map()
    return res, asg, vals


def compare(oracle, circ, ids, rows, res, asg, vals, backend):
    ores, oasg, ovals = oracle.solve_batch(oracle.Circuit(circ.to_bytes()), ids, values_from_rows(rows), len(rows), backend=backend)
    for j in range(len(rows)):
        assert res[j].as_tuple() == ores[j].as_tuple(), (j, res[j].as_tuple(), ores[j].as_tuple())
    nw = min(asg.shape[1], oasg.shape[1])
    assert np.array_equal(asg[:, :nw], oasg[:, :nw]) and np.array_equal(vals[:, :nw], ovals[:, :nw])


@pytest.mark.parametrize("force_slow", [False, True])
def test_dummy_solver_matches_reference_fake(oracle, force_slow):
    dummy = acvm_amd.make_solver(lambda pkx, pky, sig, msg: True, lambda inputs, ds: (2, 3), lambda lo, hi: (4, 5))
    circ, ids = grumpkin_circuit()
    # an arithmetic gate consuming the outputs so that the callback results flow on: w = ped_x * fixed_y + schnorr_ok
    n = circ.current_witness_index
    circ.opcodes.append(E([(1, n - 4, n - 1)], [(1, n), (P - 1, n + 1)], 0))
    circ.current_witness_index = n + 1
    rows = grumpkin_rows(40)
    res, asg, vals = solve(circ, ids, rows, dummy, force_slow)
    compare(oracle, circ, ids, rows, res, asg, vals, oracle.BACKEND_DUMMY)
    assert int.from_bytes(vals[3, n + 1].tobytes(), "big") == 2 * 5 + 1


def test_failing_and_unsupported_callbacks(oracle):
    def fixed(lo, hi):
        if lo % 3 == 0:
            raise acvm_amd.BlackBoxFailed("Limb %064x is not less than 2^128" % lo)
        if lo % 3 == 1:
            raise acvm_amd.BlackBoxUnsupported()
        return (lo, hi)
    solver = acvm_amd.make_solver(lambda *a: True, lambda inputs, ds: (0, 0), fixed)
    circ = Circuit(4, [BB("FixedBaseScalarMul", {"low": FI(1, 128), "high": FI(2, 128), "outputs": [3, 4]})])
    rows = [[j, 7] for j in range(12)]
    for slow in (False, True):
        res, asg, vals = solve(circ, [1, 2], rows, solver, slow)
        for j in range(12):
            if j % 3 == 0:
                assert (res[j].status, res[j].err, res[j].aux0) == (acvm_amd.STATUS_FAILURE, acvm_amd.ERR_BLACKBOX_FAILED, 10)
                assert res[j].message == b"Limb %064x is not less than 2^128" % j
            elif j % 3 == 1:
                assert (res[j].status, res[j].err, res[j].aux0) == (acvm_amd.STATUS_FAILURE, acvm_amd.ERR_UNSUPPORTED_BLACKBOX, 10)
            else:
                assert res[j].status == acvm_amd.STATUS_SOLVED and int.from_bytes(vals[j, 3].tobytes(), "big") == j


def test_delegating_solver_equals_builtin_kernels(oracle):
    """Callbacks that forward to the CPU restatement of barretenberg give the same witness maps as the built-in HIP backend."""
    lib = oracle.lib()

    def pedersen(inputs, ds):
        out = C.create_string_buffer(64)
        lib.oracle_pedersen(b"".join(v.to_bytes(32, "big") for v in inputs), len(inputs), ds, out)
        return int.from_bytes(out.raw[:32], "big"), int.from_bytes(out.raw[32:], "big")

    def fixed(lo, hi):
        out = C.create_string_buffer(64)
        err = C.create_string_buffer(200)
        if lib.oracle_fixed_base(lo.to_bytes(32, "big"), hi.to_bytes(32, "big"), out, err, 200) != 0:
            raise acvm_amd.BlackBoxFailed(err.value.decode())
        return int.from_bytes(out.raw[:32], "big"), int.from_bytes(out.raw[32:], "big")

    def schnorr(pkx, pky, sig, msg):
        return lib.oracle_schnorr_verify(pkx.to_bytes(32, "big") + pky.to_bytes(32, "big"), sig, len(sig), msg, len(msg)) == 1

    solver = acvm_amd.make_solver(schnorr, pedersen, fixed)
    circ, ids = grumpkin_circuit()
    rows = grumpkin_rows(24)
    res, asg, vals = solve(circ, ids, rows, solver)
    res0, asg0, vals0 = solve(circ, ids, rows, None)
    for j in range(len(rows)):
        assert res[j].as_tuple() == res0[j].as_tuple() and res[j].message == res0[j].message, j
    assert np.array_equal(asg, asg0) and np.array_equal(vals, vals0)
    compare(oracle, circ, ids, rows, res, asg, vals, oracle.BACKEND_BARRETENBERG)


def brillig_three_calls():
    """ONE Brillig program that calls all three trait functions (brillig_vm/src/black_box.rs:139-163) and a circuit that consumes the results:
    w20, w21 = fixed_base(w1, w2); w22, w23 = pedersen([w3, w4], w5); w24 = schnorr_verify(w6, w7, msg = w8..w10, sig = w11..w14);
    w25 = w20 * w23 + w24"""
    from acvm_amd.acir import Brillig
    W = E.from_witness
    bc = [("Const", 10, 100), ("BlackBox", "FixedBaseScalarMul", 0, 1, 10, 2),      # mem[100..102) = fixed_base(r0, r1)
          ("Const", 11, 2), ("Const", 12, 110), ("BlackBox", "Pedersen", 2, 11, 3, 12, 2),  # inputs: vector at r2 (the array input's pointer), size r11 = 2; domain r3
          ("Const", 13, 3), ("Const", 14, 4), ("BlackBox", "SchnorrVerify", 4, 5, 6, 13, 7, 14, 15),  # message at r6 (3 bytes), signature at r7 (4 bytes)
          ("Load", 16, 10), ("Const", 17, 101), ("Load", 17, 17), ("Load", 18, 12), ("Const", 19, 111), ("Load", 19, 19),
          ("Mov", 0, 16), ("Mov", 1, 17), ("Mov", 2, 18), ("Mov", 3, 19), ("Mov", 4, 15), ("Stop",)]
    br = Brillig(inputs=[W(1), W(2), [W(3), W(4)], W(5), W(6), W(7), [W(8), W(9), W(10)], [W(11), W(12), W(13), W(14)]], outputs=[20, 21, 22, 23, 24], bytecode=bc)
    return Circuit(25, [br, E([(1, 20, 23)], [(1, 24), (P - 1, 25)], 0)]), list(range(1, 15))


@pytest.mark.parametrize("mode", ["level", "exact", "step"])
def test_dummy_solver_inside_brillig(oracle, mode):
    """DummyBlackBoxSolver is the VM's solver too (brillig_vm/src/lib.rs:61,81,392-420): (4, 5), (2, 3), true come back into the program's
    memory and registers through the internal-call round trip, on the level schedule, on the exact kernels and one opcode at a time."""
    calls = {"schnorr": 0, "pedersen": 0, "fixed": 0}

    def schnorr(pkx, pky, sig, msg):
        calls["schnorr"] += 1
        assert len(sig) == 4 and len(msg) == 3
        return True

    def pedersen(inputs, ds):
        calls["pedersen"] += 1
        assert len(inputs) == 2
        return (2, 3)

    def fixed(lo, hi):
        calls["fixed"] += 1
        return (4, 5)
    dummy = acvm_amd.make_solver(schnorr, pedersen, fixed)
    circ, ids = brillig_three_calls()
    rows = [[j + 1, 7, 11 * j, 13, j % 3, 5, 6, 1, 2, 3, 9, 8, 7, 300 + j] for j in range(24)]
    data = circ.to_bytes()
    batch = acvm_amd.Batch(acvm_amd.Circuit(data), len(rows), ids, solver=dummy)
    batch.set_force_slow_path(mode == "exact")
    batch.set_initial_witness(values_from_rows(rows))
    if mode == "step":
        for _ in range(2):
            batch.solve_opcode()
    else:
        assert batch.solve() == 0
    res = batch.results()
    asg, vals = batch.witness_map()
    compare(oracle, circ, ids, rows, res, asg, vals, oracle.BACKEND_DUMMY)
    as_int = lambda j, w: int.from_bytes(vals[j, w].tobytes(), "big")
    assert [as_int(5, w) for w in (20, 21, 22, 23, 24, 25)] == [4, 5, 2, 3, 1, 4 * 3 + 1]
    assert calls["schnorr"] >= len(rows) and calls["pedersen"] >= len(rows) and calls["fixed"] >= len(rows)
    assert batch.get_pending_foreign_call(0) is None  # the caller never sees the internal calls
    batch.free()


def test_failing_solver_inside_brillig(oracle):
    """a callback that fails / is unsupported / panics: the VM fails at the op with BlackBoxResolutionError's Display string and the call stack
    (brillig_vm/src/lib.rs:298-307), the other instances solve; the domain separator that does not fit u32 never reaches the solver"""
    def fixed(lo, hi):
        if lo % 4 == 0:
            raise acvm_amd.BlackBoxFailed("Limb %064x is not less than 2^128" % lo)
        if lo % 4 == 1:
            raise acvm_amd.BlackBoxUnsupported()
        return (lo, hi)
    seen_ds = []

    def pedersen(inputs, ds):
        seen_ds.append(ds)
        return (2, 3)
    solver = acvm_amd.make_solver(lambda *a: True, pedersen, fixed)
    circ, ids = brillig_three_calls()
    rows = [[j, 7, 11 * j, 13, (1 << 32) if j == 6 else j % 3, 5, 6, 1, 2, 3, 9, 8, 7, 4] for j in range(12)]
    for slow in (False, True):
        batch = acvm_amd.Batch(acvm_amd.Circuit(circ.to_bytes()), len(rows), ids, solver=solver)
        batch.set_force_slow_path(slow)
        batch.set_initial_witness(values_from_rows(rows))
        batch.solve()
        res = batch.results()
        asg, vals = batch.witness_map()
        for j in range(12):
            if j % 4 == 0:
                assert (res[j].status, res[j].err, res[j].opcode_index) == (acvm_amd.STATUS_FAILURE, acvm_amd.ERR_BRILLIG_FAILED, 0), j
                assert res[j].message == b"failed to solve blackbox function: fixed_base_scalar_mul, reason: Limb %064x is not less than 2^128" % j
                assert list(res[j].call_stack[:res[j].n_call_stack]) == [1]
            elif j % 4 == 1:
                assert (res[j].status, res[j].err) == (acvm_amd.STATUS_FAILURE, acvm_amd.ERR_BRILLIG_FAILED), j
                assert res[j].message == b"unsupported blackbox function: fixed_base_scalar_mul"
            elif j == 6:
                assert (res[j].status, res[j].err) == (acvm_amd.STATUS_FAILURE, acvm_amd.ERR_BRILLIG_FAILED), j
                assert res[j].message == b"failed to solve blackbox function: pedersen, reason: Invalid signature length" and list(res[j].call_stack[:res[j].n_call_stack]) == [4]
            else:
                assert res[j].status == acvm_amd.STATUS_SOLVED and int.from_bytes(vals[j, 25].tobytes(), "big") == j * 3 + 1, j
        batch.free()
    assert (1 << 32) not in seen_ds and set(seen_ds) <= {0, 1, 2}


def test_domain_separator_is_truncated_to_128_bits_under_a_solver(oracle):
    """registers.get(domain_separator).to_u128().try_into::<u32>() (brillig_vm/src/black_box.rs:152-158): to_u128 keeps the low 128 bits
    (acir_field/src/generic_ark.rs:227-230), so 2^128 + 2 reaches the solver as separator 2 while 2^32 fails -- the same on the built-in path,
    under a caller-supplied solver (the host-side check read all 28 upper bytes until round 6) and in the oracle's VM."""
    seen_ds = []

    def pedersen(inputs, ds):
        seen_ds.append(ds)
        return (2, 3)
    solver = acvm_amd.make_solver(lambda *a: True, pedersen, lambda lo, hi: (4, 5))
    circ, ids = brillig_three_calls()
    ds_of = lambda j: [(1 << 128) + 2, 1 << 32, (1 << 200) + (1 << 128) + 1, (1 << 128) + (1 << 40), 0, 1][j % 6]
    rows = [[j + 1, 7, 11 * j, 13, ds_of(j), 5, 6, 1, 2, 3, 9, 8, 7, 4] for j in range(12)]
    for slow in (False, True):
        batch = acvm_amd.Batch(acvm_amd.Circuit(circ.to_bytes()), len(rows), ids, solver=solver)
        batch.set_force_slow_path(slow)
        batch.set_initial_witness(values_from_rows(rows))
        batch.solve()
        res = batch.results()
        asg, vals = batch.witness_map()
        compare(oracle, circ, ids, rows, res, asg, vals, oracle.BACKEND_DUMMY)
        for j in range(12):
            if j % 6 in (1, 3):
                assert res[j].message == b"failed to solve blackbox function: pedersen, reason: Invalid signature length", j
            else:
                assert res[j].status == acvm_amd.STATUS_SOLVED, (j, res[j].as_tuple(), res[j].message)
        batch.free()
    assert set(seen_ds) == {0, 1, 2}


def test_solver_with_witness_slot_reuse(oracle):
    """ACVM_BATCH_REUSE_SLOTS under a caller-supplied solver (refused until round 5): the callbacks' operands are gathered through the row map of
    the level table, the exact lanes' from their side table; results, kept witnesses and map digests against the oracle with the same backend"""
    dummy = acvm_amd.make_solver(lambda pkx, pky, sig, msg: True, lambda inputs, ds: (2, 3), lambda lo, hi: (4, 5))
    circ, ids = grumpkin_circuit()
    n = circ.current_witness_index
    circ.opcodes.append(E([(1, n - 4, n - 1)], [(1, n), (P - 1, n + 1)], 0))
    # w(n+2) = w(n+1) / w1: the instances whose w1 is zero leave the generic path here (arithmetic.rs:217-221) and are re-solved from their
    # initial witnesses by the exact kernels in the side table -- through the same callbacks
    circ.opcodes.append(E([(1, ids[0], n + 2)], [(P - 1, n + 1)], 0))
    circ.current_witness_index = n + 2
    rows = grumpkin_rows(40)
    rows[3][0] = 0
    rows[17][0] = 0
    for c, i, rows, keep in ((circ, ids, rows, [n + 1]),):
        data, values = c.to_bytes(), values_from_rows(rows)
        batch = acvm_amd.Batch(acvm_amd.Circuit(data), len(rows), i, reuse_slots=True, keep=keep, solver=dummy)
        batch.set_initial_witness(values)
        batch.solve()
        res, dig, kept = batch.results(), batch.digest(), batch.extract(keep, 0, len(rows))
        assert batch.stats()["n_slow_instances"] >= 2
        ores, oasg, ovals = oracle.solve_batch(oracle.Circuit(data), i, values, len(rows), backend=oracle.BACKEND_DUMMY)
        for j in range(len(rows)):
            assert res[j].as_tuple() == ores[j].as_tuple(), j
            assert bytes(dig[j]) == oracle.witness_map_digest(oasg[j], ovals[j]), j
            if ores[j].status == 0:
                assert bytes(kept[j][0]) == bytes(ovals[j][keep[0]]), j
        batch.free()
    # the solver INSIDE a Brillig program is a foreign-call round trip, and slot reuse takes no circuit with foreign calls (rows of an
    # instance that waits would have been recycled by the time its answer arrives): refused with the reason, not solved wrongly
    bcirc, bids = brillig_three_calls()
    with pytest.raises(acvm_amd.AcvmError, match="foreign calls"):
        acvm_amd.Batch(acvm_amd.Circuit(bcirc.to_bytes()), 4, bids, reuse_slots=True, keep=[25], solver=dummy)
