"""The reference's Brillig VM has no resource limits: memory grows on write (brillig_vm/src/memory.rs:27-39), a program runs any number of
steps at any call depth (brillig_vm/src/lib.rs:154-307). The device runs with limits (csrc/tuning.hpp) and must never turn a program
the reference solves into a failing instance: a lane that reaches a limit leaves the level schedule and the exact path retries its
opcode with the limit raised (batch.cpp retry_device_limits). Past the library's stated maxima THAT instance ends with the one outcome the
reference does not have -- Failure / ACVM_ERR_DEVICE_LIMIT, "solve it with the reference" -- and every other instance of the batch keeps its
result (the reference's caller loop loses one instance at most, acvm_js/src/execute.rs:60-119). Checked here against the oracle (which has
no such limits), through the level schedule, through the exact path, through the node driver, and one opcode at a time."""
import pytest

from acvm_amd.acir import Brillig, Circuit, Expression as E
from acvm_amd.synth import values_from_rows
from test_gpu_opcodes import both_paths, run_both

pytestmark = pytest.mark.gpu
W = E.from_witness


def store_far():
    """mem[ptr] = v at a data-dependent pointer far beyond anything the planner can estimate, read back through a second pointer"""
    bc = [("Store", 0, 1),                      # mem[r0] = r1   (memory grows with zeros up to r0)
          ("Const", 2, 1), ("BinaryFieldOp", 3, "Sub", 0, 2),
          ("Load", 4, 3),                       # r4 = mem[r0 - 1] == 0 (the zero fill)
          ("Load", 5, 0),                       # r5 = mem[r0]
          ("BinaryFieldOp", 0, "Add", 4, 5), ("Stop",)]
    return Brillig(inputs=[W(1), W(2)], outputs=[3], bytecode=bc)


def test_memory_grows_beyond_the_planner_estimate(oracle):
    circ = Circuit(4, [store_far(), E([], [(1, 3), (-1, 4)], 0)])  # w4 = w3: the opcode behind the VM still runs
    rows = [[1, 11], [63, 12], [64, 13], [200, 14], [1000, 15], [5000, 16], [70000, 17], [1, 18], [300000, 19]]
    ores, st = both_paths(oracle, circ, [1, 2], rows)
    assert all(r.status == 0 for r in ores)


def others_survive(oracle, circ, ids, rows, over, kind, **limits):
    """instance `over` is beyond a stated maximum: it alone ends with ACVM_ERR_DEVICE_LIMIT (aux0 = kind) at the Brillig opcode; every other
    instance has the oracle's result and witness map -- through the level schedule and with every instance forced through the exact path"""
    import numpy as np
    import acvm_amd
    data, values = circ.to_bytes(), values_from_rows(rows)
    ores, oasg, ovals = oracle.solve_batch(oracle.Circuit(data), ids, values, len(rows))
    assert all(r.status == 0 for r in ores)  # the reference solves all of them
    with acvm_amd.tuning(**limits):
        for force in (False, True):
            batch = acvm_amd.Batch(acvm_amd.Circuit(data), len(rows), ids)
            batch.set_initial_witness(values)
            batch.set_force_slow_path(force)
            assert batch.solve() == 1  # the call succeeds: one instance is not Solved
            res = batch.results()
            asg, vals = batch.witness_map()
            for j in range(len(rows)):
                if j == over:
                    assert (res[j].status, res[j].err, res[j].opcode_index, res[j].aux0) == (acvm_amd.STATUS_FAILURE, acvm_amd.ERR_DEVICE_LIMIT, 0, kind)
                    assert b"reference has no such limit" in res[j].message
                    assert "Not solved by this library" in batch.error_string(j)
                else:
                    assert res[j].as_tuple() == ores[j].as_tuple()
                    assert np.array_equal(asg[j], oasg[j][: asg.shape[1]]) and np.array_equal(vals[j], ovals[j][: vals.shape[1]])
            batch.free()
        return res[over]


def test_memory_beyond_the_stated_maximum_fails_the_instance_not_the_call(oracle):
    import acvm_amd
    circ = Circuit(4, [store_far(), E([], [(1, 3), (-1, 4)], 0)])
    r = others_survive(oracle, circ, [1, 2], [[5, 1], [3000, 2], [100000, 3], [7, 4]], over=2, kind=acvm_amd.LIMIT_BRILLIG_MEMORY, brillig_mem_max_log2=12)
    assert r.aux1 == 1 << 12 and b"cell 100000" in r.message
    with acvm_amd.tuning(brillig_mem_max_log2=12):  # the same handle serves the next batch
        batch = acvm_amd.Batch(acvm_amd.Circuit(circ.to_bytes()), 3, [1, 2])
        batch.set_initial_witness(values_from_rows([[5, 1], [3000, 2], [100000, 3]]))
        assert batch.solve() == 1
        batch.set_initial_witness(values_from_rows([[5, 1], [3000, 2], [4000, 3]]))
        assert batch.solve() == 0


def counting_loop():
    # r0 = n; i = 0; while i != n: i += 1; output i
    bc = [("Const", 1, 0), ("Const", 2, 1),
          ("BinaryFieldOp", 3, "Equals", 1, 0), ("JumpIf", 3, 6),
          ("BinaryFieldOp", 1, "Add", 1, 2), ("Jump", 2),
          ("Mov", 0, 1), ("Stop",)]
    return Brillig(inputs=[W(1)], outputs=[2], bytecode=bc)


def test_step_limit_is_raised_for_long_loops(oracle):
    """with the level kernels' limit lowered to 2^8 steps, loops of up to 50 000 iterations (200 000 steps) finish on the exact path"""
    import acvm_amd
    circ = Circuit(3, [counting_loop(), E([], [(1, 2), (-1, 3)], 0)])
    rows = [[n] for n in (0, 1, 10, 62, 63, 64, 65, 200, 5000, 50000)]
    with acvm_amd.tuning(brillig_steps_log2=8, brillig_steps_max_log2=20):
        ores, st = both_paths(oracle, circ, [1], rows)
    assert all(r.status == 0 for r in ores)


def test_step_limit_past_the_maximum_fails_the_instance(oracle):
    import acvm_amd
    circ = Circuit(3, [counting_loop(), E([], [(1, 2), (-1, 3)], 0)])
    r = others_survive(oracle, circ, [1], [[10], [5000], [300], [0]], over=1, kind=acvm_amd.LIMIT_BRILLIG_STEPS,  # 20 000 steps > 2^12
                       brillig_steps_log2=8, brillig_steps_max_log2=12)
    assert r.aux1 == 1 << 12


def recursion(out=2):
    # f(n): if n == 0 return; n -= 1; acc += 1; call f   -- call depth = n + 1
    bc = [("Const", 1, 0), ("Const", 2, 1), ("Const", 3, 0),   # zero, one, acc
          ("Call", 6), ("Mov", 0, 3), ("Stop",),               # 3, 4, 5
          ("BinaryFieldOp", 4, "Equals", 0, 1), ("JumpIf", 4, 11),  # 6, 7
          ("BinaryFieldOp", 0, "Sub", 0, 2), ("BinaryFieldOp", 3, "Add", 3, 2),  # 8, 9
          ("Call", 6),                                         # 10
          ("Return",)]                                         # 11
    return Brillig(inputs=[W(1)], outputs=[out], bytecode=bc)


def test_call_depth_beyond_64(oracle):
    circ = Circuit(3, [recursion(), E([], [(1, 2), (-1, 3)], 0)])
    rows = [[n] for n in (0, 1, 62, 63, 64, 65, 66, 300, 2000)]
    ores, st = both_paths(oracle, circ, [1], rows)
    assert all(r.status == 0 for r in ores)


def test_call_depth_past_the_maximum_fails_the_instance(oracle):
    import acvm_amd
    circ = Circuit(3, [recursion(), E([], [(1, 2), (-1, 3)], 0)])
    r = others_survive(oracle, circ, [1], [[100], [500], [3], [90]], over=1, kind=acvm_amd.LIMIT_BRILLIG_CALL_DEPTH, brillig_call_depth_max=128)
    assert r.aux1 == 128


def test_node_driver_returns_the_other_instances(oracle):
    """acvm_node_solve over tiles with the exact path beside the next tile: the instance past a maximum comes back as ACVM_ERR_DEVICE_LIMIT in the
    caller's arrays, the 199 others as the oracle has them"""
    import numpy as np
    import acvm_amd
    circ = Circuit(4, [store_far(), E([], [(1, 3), (-1, 4)], 0)])
    rows = [[5 + (j % 40), j + 1] for j in range(200)]
    rows[77] = [100000, 9]
    data, values = circ.to_bytes(), values_from_rows(rows)
    ores, oasg, ovals = oracle.solve_batch(oracle.Circuit(data), [1, 2], values, len(rows))
    with acvm_amd.tuning(brillig_mem_max_log2=12):
        node = acvm_amd.Node(acvm_amd.Circuit(data), [1, 2], keep=[4], devices=[0, 0], tile=64)
        not_solved, res, kept, asg, dig = node.solve(values, len(rows))
        node.free()
    assert not_solved == 1 and (res[77].status, res[77].err, res[77].aux0) == (acvm_amd.STATUS_FAILURE, acvm_amd.ERR_DEVICE_LIMIT, acvm_amd.LIMIT_BRILLIG_MEMORY)
    for j in range(len(rows)):
        if j != 77:
            assert res[j].as_tuple() == ores[j].as_tuple() and asg[j, 0] == 1 and bytes(kept[j, 0]) == bytes(ovals[j][4])
            assert bytes(dig[j]) == oracle.witness_map_digest(oasg[j], ovals[j])


def test_a_failing_program_still_fails_after_a_retry(oracle):
    """a lane that needs more memory AND then traps: the retry must report the trap with its call stack, like the oracle"""
    bc = [("Store", 0, 1), ("Call", 3), ("Stop",), ("Trap",)]
    circ = Circuit(3, [Brillig(inputs=[W(1), W(2)], outputs=[3], bytecode=bc)])
    ores, _ = both_paths(oracle, circ, [1, 2], [[3, 1], [9000, 2]])
    assert ores[1].err == oracle.E_BRILLIG_FAILED and list(ores[1].call_stack[:ores[1].n_call_stack]) == [1, 3]


def test_limits_while_stepping(oracle):
    """acvm_batch_solve_opcode (ACVM::solve_opcode, pwg/mod.rs:243-303): the retry happens inside the step that executes the opcode"""
    import acvm_amd
    circ = Circuit(5, [store_far(), recursion(out=4), E([], [(1, 3), (1, 4), (-1, 5)], 0)])
    # recursion reads w1 as n and writes w4; store_far reads (w1, w2) and writes w3
    rows = [[70, 5], [3000, 6], [2, 7]]
    data = circ.to_bytes()
    batch = acvm_amd.Batch(acvm_amd.Circuit(data), len(rows), [1, 2])
    batch.set_initial_witness(values_from_rows(rows))
    for _ in range(3):
        batch.solve_opcode()
    assert all(r.status == 0 for r in batch.results())
    asg, vals = batch.witness_map()
    ores, oasg, ovals = oracle.solve_batch(oracle.Circuit(data), [1, 2], values_from_rows(rows), len(rows))
    assert (asg == oasg[:, :asg.shape[1]]).all() and (vals == ovals[:, :vals.shape[1]]).all()
    assert batch.stats()["n_brillig_retries"] >= 1


def test_step_limit_past_the_maximum_while_stepping(oracle):
    """the instance that is given up stays ACVM_ERR_DEVICE_LIMIT on every later acvm_batch_solve_opcode, is not retried again (the retry
    counter stands still), and the other instances step on to the oracle's map -- also when the given-up lane is the last one running"""
    import acvm_amd
    circ = Circuit(4, [counting_loop(), E([], [(1, 2), (-1, 3)], 0), E([], [(1, 3), (-1, 4)], 0)])
    for rows, over in (([[10], [5000], [300]], 1), ([[5000]], 0)):
        data, values = circ.to_bytes(), values_from_rows(rows)
        ores, oasg, ovals = oracle.solve_batch(oracle.Circuit(data), [1], values, len(rows))
        with acvm_amd.tuning(brillig_steps_log2=8, brillig_steps_max_log2=12):
            batch = acvm_amd.Batch(acvm_amd.Circuit(data), len(rows), [1])
            batch.set_initial_witness(values)
            batch.solve_opcode()
            retries = batch.stats()["n_brillig_retries"]
            assert retries >= 1
            for step in range(3):
                res = batch.results()
                assert (res[over].status, res[over].err, res[over].opcode_index, res[over].aux0) == (acvm_amd.STATUS_FAILURE, acvm_amd.ERR_DEVICE_LIMIT, 0, acvm_amd.LIMIT_BRILLIG_STEPS), step
                batch.solve_opcode()
                assert batch.stats()["n_brillig_retries"] == retries, step
            res = batch.results()
            asg, vals = batch.witness_map()
            for j in range(len(rows)):
                if j != over:
                    assert res[j].as_tuple() == ores[j].as_tuple()
                    assert (asg[j] == oasg[j][: asg.shape[1]]).all() and (vals[j] == ovals[j][: vals.shape[1]]).all()
            batch.free()
