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
