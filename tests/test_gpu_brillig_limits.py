"""The reference's Brillig VM has no resource limits: memory grows on write (brillig_vm/src/memory.rs:27-39), a program runs any number of
steps at any call depth (brillig_vm/src/lib.rs:154-307). The device runs with limits (csrc/tuning.hpp) and must never turn a program
the reference solves into a failing instance: a lane that reaches a limit leaves the level schedule and the exact path retries its
opcode with the limit raised (batch.cpp retry_device_limits); only past the library's stated maxima does the SOLVE CALL fail
(ACVM_E_UNSUPPORTED), never the instance. Checked here against the oracle (which has no such limits), through the level schedule,
through the exact path, and one opcode at a time."""
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


def test_memory_beyond_the_stated_maximum_fails_the_solve_call_not_the_instance(oracle):
    import acvm_amd
    circ = Circuit(3, [store_far()])
    with acvm_amd.tuning(brillig_mem_max_log2=12):
        batch = acvm_amd.Batch(acvm_amd.Circuit(circ.to_bytes()), 3, [1, 2])
        batch.set_initial_witness(values_from_rows([[5, 1], [3000, 2], [100000, 3]]))  # the last one needs more than 2^12 cells
        with pytest.raises(acvm_amd.AcvmError, match="error -3.*instance 2.*memory cell 100000"):
            batch.solve()
        with pytest.raises(acvm_amd.AcvmError, match="not solved"):
            batch.results()
        # the same handle, inputs the device can hold: solves
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


def test_step_limit_past_the_maximum_fails_the_solve_call(oracle):
    import acvm_amd
    circ = Circuit(2, [counting_loop()])
    with acvm_amd.tuning(brillig_steps_log2=8, brillig_steps_max_log2=12):
        batch = acvm_amd.Batch(acvm_amd.Circuit(circ.to_bytes()), 2, [1])
        batch.set_initial_witness(values_from_rows([[10], [5000]]))  # 20 000 steps > 2^12
        with pytest.raises(acvm_amd.AcvmError, match="error -3.*instance 1.*VM steps"):
            batch.solve()


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


def test_call_depth_past_the_maximum_fails_the_solve_call(oracle):
    import acvm_amd
    circ = Circuit(2, [recursion()])
    with acvm_amd.tuning(brillig_call_depth_max=128):
        batch = acvm_amd.Batch(acvm_amd.Circuit(circ.to_bytes()), 2, [1])
        batch.set_initial_witness(values_from_rows([[100], [500]]))
        with pytest.raises(acvm_amd.AcvmError, match="error -3.*instance 1.*calls"):
            batch.solve()


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
