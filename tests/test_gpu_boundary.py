"""The rest of the drop-in boundary on the device: ACVM::solve_opcode stepping for a batch, the single-instance
shim acvm_new / acvm_solve / acvm_solve_opcode / acvm_finalize (struct ACVM, acvm/src/pwg/mod.rs:129-304), instances with
different initial-witness id sets (acvm_multi_*), the library's built-in StubbedBackend / DummyBlackBoxSolver vtables and the
batched members of the BlackBoxFunctionSolver vtable -- each against the CPU oracle driven the same way."""
import numpy as np
import pytest

import acvm_amd
from acvm_amd import synth
from acvm_amd.acir import P, BlackBoxFuncCall as BB, Brillig, Circuit, Expression as E, FunctionInput as FI
from acvm_amd.synth import grumpkin_circuit, grumpkin_rows, values_from_rows

pytestmark = pytest.mark.gpu
W = E.from_witness


def _oracle_maps(oracle, data, ids, values, B, backend=0):
    return oracle.solve_batch(oracle.Circuit(data), ids, values, B, backend=backend)


def test_batch_solve_opcode_steps_like_the_reference(oracle):
    """one acvm_batch_solve_opcode = one ACVM::solve_opcode of every instance: status, instruction pointer and the partial
    witness map after EVERY step equal the oracle's, failing instances stop where the reference stops"""
    circ, ids = synth.mixed_circuit(60, seed=0xAC1D0B01)
    data = circ.to_bytes()
    B = 12
    values = synth.witness_batch(B, seed=0xAC1D0B01)
    row = len(ids) * 32
    batch = acvm_amd.Batch(acvm_amd.Circuit(data), B, ids)
    batch.set_initial_witness(values)
    oc = oracle.Circuit(data)
    refs = [oracle.ACVM(oc, {w: int.from_bytes(values[j * row + 32 * k:j * row + 32 * k + 32], "big") for k, w in enumerate(ids)}) for j in range(B)]
    n_ops = len(circ.opcodes)
    for step in range(n_ops):
        left = batch.solve_opcode()
        for a in refs:
            if a.result().status == oracle.ST_IN_PROGRESS:
                a.solve_opcode()
        res = batch.results()
        for j, a in enumerate(refs):
            r = a.result()
            assert res[j].status == r.status, (step, j)
            if r.status == oracle.ST_IN_PROGRESS:
                assert res[j].opcode_index == a.instruction_pointer() == step + 1
            elif r.status == oracle.ST_FAILURE:
                assert res[j].as_tuple() == r.as_tuple()
        assert left == sum(1 for a in refs if a.result().status != oracle.ST_SOLVED)
        if step in (0, 7, n_ops // 2, n_ops - 1):
            asg, vals = batch.witness_map()
            for j, a in enumerate(refs):
                got = {w: int.from_bytes(vals[j, w].tobytes(), "big") for w in range(asg.shape[1]) if asg[j, w]}
                assert got == a.witness_map(), (step, j)
    assert all(a.result().status in (oracle.ST_SOLVED, oracle.ST_FAILURE) for a in refs)
    assert batch.solve_opcode() == left  # nothing is left to step
    # solve() after a few steps runs the rest; a plain solve() gives the same final maps
    b2 = acvm_amd.Batch(acvm_amd.Circuit(data), B, ids)
    b2.set_initial_witness(values)
    for _ in range(5):
        b2.solve_opcode()
    b2.solve()
    b3 = acvm_amd.Batch(acvm_amd.Circuit(data), B, ids)
    b3.set_initial_witness(values)
    b3.solve()
    for x in (b2, b3):
        assert [r.as_tuple() for r in x.results()] == [r.as_tuple() for r in batch.results()]
        a2, v2 = x.witness_map()
        a1, v1 = batch.witness_map()
        assert np.array_equal(a1, a2) and np.array_equal(v1, v2)
    with pytest.raises(acvm_amd.AcvmError):
        b3.solve_opcode()  # stepping starts from a fresh batch


def test_single_instance_shim_addition(golden, oracle):
    fx = golden["acvm_js"]["addition"]
    iw = {int(k): int(v, 16) for k, v in fx["initialWitnessMap"].items()}
    a = acvm_amd.Acvm(acvm_amd.Circuit(bytes(fx["bytecode"])), iw)
    assert a.status().status == acvm_amd.STATUS_IN_PROGRESS and a.instruction_pointer() == 0
    assert a.witness_map() == iw
    with pytest.raises(acvm_amd.AcvmError):
        a.finalize()  # "ACVM is not ready to be finalized"
    assert a.solve() == acvm_amd.STATUS_SOLVED
    full = a.finalize()
    assert {w: full[w] for w in iw} == iw and full[int(fx["resultWitness"])] == int(fx["expectedResult"], 16)
    ref = oracle.ACVM(oracle.Circuit(bytes(fx["bytecode"])), iw)
    ref.solve()
    assert full == ref.witness_map()


def test_single_instance_shim_steps_and_foreign_call(golden, oracle):
    fx = golden["acvm_js"]["foreign_call"]
    iw = {int(k): int(v, 16) for k, v in fx["initialWitnessMap"].items()}
    data = bytes(fx["bytecode"])
    a = acvm_amd.Acvm(acvm_amd.Circuit(data), iw)
    ref = oracle.ACVM(oracle.Circuit(data), iw)
    for _ in range(64):
        st, rst = a.solve_opcode(), ref.solve_opcode()
        assert st == rst and a.instruction_pointer() == ref.instruction_pointer()
        if st == acvm_amd.STATUS_REQUIRES_FOREIGN_CALL:
            fn, inputs = a.get_pending_foreign_call()
            assert (fn, inputs) == ref.get_pending_foreign_call()
            resp = [int(x, 16) if isinstance(x, str) else [int(y, 16) for y in x] for x in fx["oracleResponse"]]
            a.resolve_pending_foreign_call(resp)
            ref.resolve_pending_foreign_call(resp)
        assert a.witness_map() == ref.witness_map()
        if st == acvm_amd.STATUS_SOLVED:
            break
    assert a.finalize() == {int(k): int(v, 16) for k, v in fx["expectedWitnessMap"].items()}


def test_instances_with_different_initial_sets(oracle):
    """w3 = w1 * w2; w4 = w3 + w1. Instances give {1, 2}, {1, 2, 3} (consistent or not), {2, 3} (w1 unknown: solved from the
    product), {1} (unsolvable): each against its own oracle ACVM"""
    ops = [E([(1, 1, 2)], [(P - 1, 3)], 0), E([], [(1, 3), (1, 1), (P - 1, 4)], 0)]
    circ = Circuit(4, ops)
    data = circ.to_bytes()
    maps = [{1: 3, 2: 5}, {2: 5, 1: 3, 3: 15}, {1: 3, 2: 5, 3: 16}, {2: 5, 3: 15}, {1: 7}, {1: 4, 2: 6}, {3: 15, 2: 5}]
    m = acvm_amd.MultiBatch(acvm_amd.Circuit(data), maps)
    assert m.n_groups == 4
    m.solve()
    res = m.results()
    oc = oracle.Circuit(data)
    for i, iw in enumerate(maps):
        a = oracle.ACVM(oc, iw)
        a.solve()
        assert res[i].as_tuple() == a.result().as_tuple(), (i, res[i].as_tuple(), a.result().as_tuple())
        asg, vals = m.witness_map(i)
        got = {w: int.from_bytes(vals[w].tobytes(), "big") for w in range(len(asg)) if asg[w]}
        assert got == a.witness_map(), i


@pytest.mark.parametrize("force_slow", [False, True])
def test_builtin_dummy_and_stubbed_vtables(oracle, force_slow):
    circ, ids = grumpkin_circuit()
    rows = grumpkin_rows(20)
    values = values_from_rows(rows)
    for solver, backend in ((acvm_amd.bb_dummy(), oracle.BACKEND_DUMMY), (acvm_amd.bb_stubbed(), oracle.BACKEND_STUBBED)):
        batch = acvm_amd.Batch(acvm_amd.Circuit(circ.to_bytes()), len(rows), ids, solver=solver)
        batch.set_force_slow_path(force_slow)
        batch.set_initial_witness(values)
        batch.solve()
        res = batch.results()
        asg, vals = batch.witness_map()
        ores, oasg, ovals = _oracle_maps(oracle, circ.to_bytes(), ids, values, len(rows), backend)
        for j in range(len(rows)):
            assert res[j].as_tuple() == ores[j].as_tuple() and res[j].message == ores[j].message, (j, res[j].message, ores[j].message)
        assert np.array_equal(asg, oasg[:, :asg.shape[1]]) and np.array_equal(vals, ovals[:, :vals.shape[1]])
    assert res[0].err == acvm_amd.ERR_PANIC and res[0].message == b"Path not trodden by this test"


def test_batched_vtable_members_are_called_once_per_opcode(oracle):
    calls = {"ped": 0, "fixed": 0, "schnorr": 0}
    lib = oracle.lib()
    import ctypes as C

    def ped(rows, ds):
        calls["ped"] += 1
        out = []
        for r in rows:
            buf = C.create_string_buffer(64)
            lib.oracle_pedersen(b"".join(v.to_bytes(32, "big") for v in r), len(r), ds, buf)
            out.append((int.from_bytes(buf.raw[:32], "big"), int.from_bytes(buf.raw[32:], "big")))
        return out

    def fixed(pairs):
        calls["fixed"] += 1
        return [(lo % P, hi % P) for lo, hi in pairs]

    def schnorr(items):
        calls["schnorr"] += 1
        return [lib.oracle_schnorr_verify(x.to_bytes(32, "big") + y.to_bytes(32, "big"), sig, len(sig), msg, len(msg)) == 1 for x, y, sig, msg in items]

    solver = acvm_amd.make_batched_solver(ped, fixed, schnorr)
    circ, ids = grumpkin_circuit()
    rows = grumpkin_rows(300)
    batch = acvm_amd.Batch(acvm_amd.Circuit(circ.to_bytes()), len(rows), ids, solver=solver)
    batch.set_initial_witness(values_from_rows(rows))
    batch.solve()
    assert calls == {"ped": 1, "fixed": 1, "schnorr": 1}
    res = batch.results()
    asg, vals = batch.witness_map()
    # Pedersen and Schnorr as the built-in backend computes them; the fixed-base fake echoes its inputs
    b0 = acvm_amd.Batch(acvm_amd.Circuit(circ.to_bytes()), len(rows), ids)
    b0.set_initial_witness(values_from_rows(rows))
    b0.solve()
    asg0, vals0 = b0.witness_map()
    n_in = len(ids)
    assert np.array_equal(vals[:, n_in + 1:n_in + 3], vals0[:, n_in + 1:n_in + 3])  # Pedersen x, y
    # SchnorrVerify (instances 0..7 stop at the built-in backend's limb / modulus checks of FixedBaseScalarMul, which the fake does not make)
    assert np.array_equal(vals[8:, n_in + 5], vals0[8:, n_in + 5]) and vals[:, n_in + 5, 31].sum() > 100
    for j, r in enumerate(rows):
        assert res[j].status == acvm_amd.STATUS_SOLVED
        assert int.from_bytes(vals[j, n_in + 3].tobytes(), "big") == r[2] % P and int.from_bytes(vals[j, n_in + 4].tobytes(), "big") == r[3] % P


def test_multi_repeated_id_keeps_its_last_value(oracle):
    """a map has every id once: an instance that lists an id twice keeps the LAST value, like BTreeMap::insert (acvm_multi_new used to refuse it)"""
    import ctypes as C
    import numpy as np
    ops = [E([(1, 1, 2)], [(P - 1, 3)], 0)]
    circ = Circuit(3, ops)
    gc = acvm_amd.Circuit(circ.to_bytes())
    ids = np.asarray([1, 2, 1, 1, 2], dtype=np.uint32)          # instance 0: {1: 3 -> 9, 2: 5}; instance 1: {1: 4, 2: 6}
    offsets = np.asarray([0, 3, 5], dtype=np.uint64)
    vals = b"".join(int(v).to_bytes(32, "big") for v in (3, 5, 9, 4, 6))
    L = acvm_amd.lib()
    h = L.acvm_multi_new(gc._h, None, 2, offsets.ctypes.data, ids.ctypes.data, vals)
    assert h, L.acvm_last_error()
    assert L.acvm_multi_num_groups(h) == 1 and L.acvm_multi_solve(h) == 0
    nw = L.acvm_multi_num_witnesses(h)
    for inst, want in ((0, 45), (1, 24)):
        asg = np.zeros(nw, dtype=np.uint8)
        v = np.zeros((nw, 32), dtype=np.uint8)
        assert L.acvm_multi_witness_map(h, inst, asg.ctypes.data, v.ctypes.data) == 0
        assert asg[3] and int.from_bytes(v[3].tobytes(), "big") == want
    L.acvm_multi_free(h)


@pytest.mark.parametrize("flags", [{}, {"fold_digest": True, "reuse_slots": True}])
def test_one_handle_serves_smaller_batches(oracle, flags):
    """acvm_batch_set_instances: a handle created for 300 instances solves batches of 300, 70, 1 and 300 again -- results, every witness and the
    digests equal the oracle's for exactly the live instances (edge-case inputs: some take the exact path)"""
    circ, ids = synth.mixed_circuit(400, seed=0xAC1D0B07)
    data = circ.to_bytes()
    gc = acvm_amd.Circuit(data)
    keep = gc.witness_set("return_values") + [ids[0]]
    try:
        batch = acvm_amd.Batch(gc, 300, ids, keep=keep, **flags)
    except acvm_amd.AcvmError as e:
        pytest.skip(f"planner refuses this mode for the circuit: {e}")
    oc = oracle.Circuit(data)
    for n, seed in ((300, 1), (70, 2), (1, 3), (300, 4)):
        values = synth.witness_batch(n, seed=0xAC1D0B07 + seed, edge_cases=True)
        batch.set_instances(n)
        batch.set_initial_witness(values)
        batch.solve()
        res = batch.results()
        ores, oasg, ovals = oracle.solve_batch(oc, ids, values, n)
        assert len(res) == n and [r.as_tuple() for r in res] == [r.as_tuple() for r in ores]
        dig = batch.digest()
        assert dig.shape[0] == n
        for j in range(n):
            assert bytes(dig[j]) == oracle.witness_map_digest(oasg[j], ovals[j]), (n, j)
        for w in keep:
            v, a = batch.witness(w)
            assert v.shape[0] == n and np.array_equal(a.astype(bool), oasg[:, w].astype(bool)) and np.array_equal(v[a.astype(bool)], ovals[:, w][a.astype(bool)])
    with pytest.raises(acvm_amd.AcvmError):
        batch.set_instances(301)
    with pytest.raises(acvm_amd.AcvmError):
        batch.set_instances(0)
    batch.free()
