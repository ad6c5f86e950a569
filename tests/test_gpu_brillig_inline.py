"""Straight-line Brillig (plan.cpp emit_straight_line, ops_light.hpp op_brillig_sl): Brillig opcodes without loops, calls and memory run in
the level schedule as light records instead of in the VM kernel. Every shape is compared with the oracle's VM (brillig_vm/src/lib.rs,
arithmetic.rs:7-81, acvm/src/pwg/brillig.rs:20-150) through the level schedule (inlined), with the inlining switched off (the VM
kernel), and through the exact path (always the VM)."""
import random

import pytest

from acvm_amd.acir import P, Brillig, Circuit, Expression as E
from test_gpu_opcodes import both_paths, run_both

pytestmark = pytest.mark.gpu
W = E.from_witness


def check(oracle, circ, ids, rows, inlined):
    import acvm_amd
    ores, st = both_paths(oracle, circ, ids, rows)
    assert st["n_brillig_inlined"] == inlined, st["n_brillig_inlined"]
    with acvm_amd.tuning(brillig_inline=0):
        _, st0 = run_both(oracle, circ, ids, rows)
    assert st0["n_brillig_inlined"] == 0
    return ores


def test_stdlib_integer_shapes(oracle):
    """stdlib/src/blackbox_fallbacks/uint.rs:212-260: r0 op= r1 at a fixed width, with and without a guard constant"""
    r = random.Random(1)
    ops = []
    out = 3
    for op in ("Add", "Sub", "Mul", "UnsignedDiv", "SignedDiv", "And", "Or", "Xor", "Equals", "LessThan", "LessThanEquals", "Shl", "Shr"):
        for bits in (8, 32, 64, 127):
            ops.append(Brillig(inputs=[W(1), W(2)], outputs=[out], bytecode=[("BinaryIntOp", 0, op, bits, 0, 1), ("Stop",)]))
            out += 1
            ops.append(Brillig(inputs=[W(1), W(2)], outputs=[out], bytecode=[("Const", 2, 1), ("BinaryIntOp", 1, "Add", bits, 1, 2), ("BinaryIntOp", 0, op, bits, 0, 1)]))
            out += 1
    circ = Circuit(out, ops)
    # small second operands so that shifts and divisions mostly succeed; every panic shape appears in some instance and flags it
    rows = [[r.randrange(1 << 64), r.randrange(1, 64)] for _ in range(40)] + [[r.randrange(P), r.randrange(P)] for _ in range(12)] + [[5, 0], [0, 0], [P - 1, 1]]
    ores = check(oracle, circ, [1, 2], rows, inlined=len(ops))
    assert sum(1 for x in ores if x.status == 0) >= 20 and any(x.status == 2 for x in ores)


def test_forward_jumps_inversion_hint_and_register_remap(oracle):
    # the compiler's inversion hint: if x == 0 skip, else x = 1 / x (registers 0, 7: remapped into the four slots)
    inv = Brillig(inputs=[W(1)], outputs=[10], bytecode=[("JumpIfNot", 0, 3), ("Const", 7, 1), ("BinaryFieldOp", 0, "Div", 7, 0), ("Stop",)])
    # select: out = c ? a : b with a jump over a jump, code behind a Stop, and a jump past the end (= finished)
    sel = Brillig(inputs=[W(1), W(2), W(3)], outputs=[11], bytecode=[("JumpIf", 2, 3), ("Mov", 0, 1), ("Stop",), ("Jump", 99), ("Mov", 0, 2)])
    # comparison chain with two outputs and a register far away
    cmp2 = Brillig(inputs=[W(1), W(2)], outputs=[12, 13], bytecode=[("BinaryIntOp", 300, "LessThan", 64, 0, 1), ("BinaryFieldOp", 1, "Mul", 0, 1),
                                                                     ("Mov", 0, 300), ("JumpIf", 0, 5), ("Const", 1, 77), ("Stop",)])
    trap = Brillig(inputs=[W(3)], outputs=[14], bytecode=[("JumpIfNot", 0, 2), ("Trap",), ("Const", 0, 5), ("Stop",)])  # traps where w3 != 0
    pred = Brillig(inputs=[W(1), W(2)], outputs=[15], bytecode=[("BinaryFieldOp", 0, "Add", 0, 1), ("Stop",)], predicate=W(3))  # predicate 0: outputs := 0
    circ = Circuit(15, [inv, sel, cmp2, pred, trap])
    r = random.Random(2)
    rows = [[r.randrange(P), r.randrange(P), r.randrange(2)] for _ in range(60)] + [[0, 0, 0], [0, 5, 1], [1, 0, 0], [P - 1, P - 1, 1], [3, 1 << 70, 0]]
    ores = check(oracle, circ, [1, 2, 3], rows, inlined=5)
    assert any(x.status == 2 and x.opcode_index == 4 for x in ores) and any(x.status == 0 for x in ores)


def test_output_conflicts_and_unknown_inputs(oracle):
    """an output that is already assigned is compared (insert_value, pwg/mod.rs:338-357); an input expression over an unassigned witness is
    ExpressionHasTooManyUnknowns for every instance (the plan is truncated there)"""
    a = Brillig(inputs=[W(1), W(2)], outputs=[2], bytecode=[("BinaryFieldOp", 0, "Add", 0, 1), ("Stop",)])      # w2 is an input: conflict unless w1 == 0
    b = Brillig(inputs=[W(1)], outputs=[4, 4], bytecode=[("Mov", 1, 0), ("Stop",)])                               # the same witness twice: second is compared
    c = Brillig(inputs=[E([(3, 1, 2)], [(2, 1)], 7)], outputs=[5], bytecode=[("Const", 1, 2), ("BinaryFieldOp", 0, "Mul", 0, 1)])  # expression input
    d = Brillig(inputs=[W(9)], outputs=[6], bytecode=[("Stop",)])                                                  # w9 never assigned
    circ = Circuit(9, [a, b, c, d])
    rows = [[0, 5], [1, 5], [0, 0], [P - 1, 1]]
    ores = check(oracle, circ, [1, 2], rows, inlined=3)  # d is never reached by the replay: the plan stops at it
    assert ores[0].err == oracle.E_TOO_MANY_UNKNOWNS and ores[0].opcode_index == 3 and ores[1].err == oracle.E_UNSATISFIED


def test_not_inlined_shapes_keep_the_vm(oracle):
    loop = Brillig(inputs=[W(1)], outputs=[3], bytecode=[("Const", 1, 1), ("BinaryFieldOp", 0, "Sub", 0, 1), ("JumpIf", 0, 1), ("Stop",)])   # backward jump
    mem = Brillig(inputs=[W(1)], outputs=[4], bytecode=[("Const", 1, 0), ("Store", 1, 0), ("Load", 0, 1), ("Stop",)])
    arr = Brillig(inputs=[[W(1), W(2)]], outputs=[5], bytecode=[("Load", 0, 0), ("Stop",)])
    regs = Brillig(inputs=[W(1)], outputs=[6], bytecode=[("Mov", 10, 0), ("Mov", 11, 10), ("Mov", 12, 11), ("Mov", 13, 12), ("Mov", 0, 13)])  # five registers
    wide = Brillig(inputs=[W(1), W(2)], outputs=[7], bytecode=[("BinaryIntOp", 0, "And", 256, 0, 1), ("Stop",)])
    circ = Circuit(7, [loop, mem, arr, regs, wide])
    rows = [[k, 3 * k + 1] for k in range(1, 20)]
    check(oracle, circ, [1, 2], rows, inlined=1)
