"""GPU parity of Opcode::Brillig (the Brillig VM on the device) against the CPU oracle: field / integer ALU incl. the
known answers of brillig_vm/src/arithmetic.rs:149-234, control flow (per-lane divergent loops, call / return, trap),
memory, predicates, array inputs / outputs, black-box ops (brillig_vm/src/black_box.rs), failure shapes."""
import random

import pytest

from acvm_amd.acir import P, BINARY_INT_OPS, Brillig, Circuit, Expression as E
from test_gpu_opcodes import both_paths, run_both

pytestmark = pytest.mark.gpu
W = E.from_witness


def edge_values(bits):
    return [0, 1, 2, (1 << bits) - 1 if bits < 254 else P - 1, 1 << (bits - 1) if 0 < bits < 254 else 5, (1 << bits) % P if bits < 254 else 7,
            P - 1, P - 2, (1 << 128) - 1, 1 << 128, 255, 256]


def int_rows(r, bits, n):
    ev = edge_values(bits)
    rows = [[a, b] for a in ev[:6] for b in ev[:6]]
    while len(rows) < n:
        k = r.randrange(4)
        a = r.randrange(P) if k == 0 else r.randrange(1 << max(bits, 1))
        b = r.randrange(P) if k == 1 else r.randrange(1 << max(bits, 1))
        rows.append([a % P, b % P])
    return rows


@pytest.mark.parametrize("bits", [1, 4, 8, 32, 64, 127, 128, 200, 254, 256])
def test_int_ops_never_panicking(oracle, bits):
    """Add, Mul, Equals, LessThan, LessThanEquals, And, Or, Xor (+ shifts for bit_size <= 128), one Brillig opcode each."""
    r = random.Random(bits)
    ops = ["Add", "Mul", "Equals", "LessThan", "LessThanEquals", "And", "Or", "Xor"]
    opcodes = [Brillig(inputs=[W(1), W(2)], outputs=[3 + i], bytecode=[("BinaryIntOp", 0, op, bits, 0, 1), ("Stop",)]) for i, op in enumerate(ops)]
    # field ops in one program: r2 = a+b, r3 = a-b, r4 = a*b, r5 = (a == b); outputs are registers 0..3 after the moves
    opcodes.append(Brillig(inputs=[W(1), W(2)], outputs=[20, 21, 22, 23],
                           bytecode=[("BinaryFieldOp", 2, "Add", 0, 1), ("BinaryFieldOp", 3, "Sub", 0, 1), ("BinaryFieldOp", 4, "Mul", 0, 1),
                                     ("BinaryFieldOp", 5, "Equals", 0, 1), ("Mov", 0, 2), ("Mov", 1, 3), ("Mov", 2, 4), ("Mov", 3, 5), ("Stop",)]))
    opcodes.append(Brillig(inputs=[W(1), W(2)], outputs=[24], bytecode=[("BinaryFieldOp", 0, "Div", 0, 1)]))  # inverse(0) == 0
    both_paths(oracle, Circuit(30, opcodes), [1, 2], int_rows(r, bits, 70))


@pytest.mark.parametrize("op,bits", [("Sub", 4), ("Sub", 64), ("Sub", 254), ("Sub", 256), ("UnsignedDiv", 8), ("UnsignedDiv", 64), ("UnsignedDiv", 254),
                                     ("SignedDiv", 8), ("SignedDiv", 32), ("SignedDiv", 127), ("SignedDiv", 256), ("SignedDiv", 0),
                                     ("Shl", 8), ("Shl", 64), ("Shl", 128), ("Shr", 8), ("Shr", 128), ("Shl", 129), ("Shr", 200), ("Shl", 300)])
def test_int_ops_that_can_panic(oracle, op, bits):
    r = random.Random(hash((op, bits)) & 0xFFFF)
    circ = Circuit(3, [Brillig(inputs=[W(1), W(2)], outputs=[3], bytecode=[("BinaryIntOp", 0, op, bits, 0, 1), ("Stop",)])])
    rows = int_rows(r, min(bits, 254), 80)
    if op in ("Shl", "Shr"):
        rows += [[r.randrange(P), s] for s in (0, 1, 7, 8, 31, 32, 33, 63, 64, 65, 127, 128, 255, 256, 257, 1 << 64, 1 << 130)]
    both_paths(oracle, circ, [1, 2], rows)


@pytest.mark.parametrize("bits", [257, 300, 507, 508, 512, 1000, (1 << 20) + 3, (1 << 32) - 1])
def test_int_ops_beyond_256_bits(oracle, bits):
    """the reference's BigUint arithmetic takes any bit_size (brillig_vm/src/arithmetic.rs:23-34): Sub wraps at 2^bits (a residue mod p the
    planner precomputes), Mul masks its 512-bit product, everything else no longer sees the modulus; one opcode per op, inlined and in the VM"""
    import acvm_amd
    r = random.Random(bits)
    ops = ["Add", "Sub", "Mul", "UnsignedDiv", "SignedDiv", "Equals", "LessThan", "LessThanEquals", "And", "Or", "Xor"]
    opcodes = [Brillig(inputs=[W(1), W(2)], outputs=[3 + i], bytecode=[("BinaryIntOp", 0, op, bits, 0, 1), ("Stop",)]) for i, op in enumerate(ops)]
    # the same through the VM kernel (a backward jump keeps the program out of the straight-line records)
    opcodes += [Brillig(inputs=[W(1), W(2)], outputs=[20 + i], bytecode=[("BinaryIntOp", 0, op, bits, 0, 1), ("Const", 1, 0), ("JumpIf", 1, 0), ("Stop",)]) for i, op in enumerate(ops)]
    ev = [1, 2, P - 1, P - 2, (1 << 253) + 12345, (1 << 128) - 1, 1 << 200]
    rows = [[a, b] for a in ev for b in ev] + [[r.randrange(P), r.randrange(1, P)] for _ in range(40)]
    ores, st = both_paths(oracle, Circuit(40, opcodes), [1, 2], rows)
    assert st["n_brillig_inlined"] == len(ops) and all(x.status == 0 for x in ores)
    both_paths(oracle, Circuit(3, [Brillig(inputs=[W(1), W(2)], outputs=[3], bytecode=[("BinaryIntOp", 0, "UnsignedDiv", bits, 0, 1)])]), [1, 2], [[5, 0], [0, 0], [7, 2]])


def test_int_known_answers(oracle):
    """brillig_vm/src/arithmetic.rs:149-234, checked on the device against literal expected values."""
    import acvm_amd
    from acvm_amd.synth import values_from_rows
    neg = lambda x, bits: (1 << bits) - x  # noqa: E731
    cases = [("Add", 4, 5, 10, 15), ("Add", 4, 10, 10, 4), ("Add", 4, 5, neg(3, 4), 2), ("Sub", 4, 5, 3, 2), ("Sub", 4, 5, 10, neg(5, 4)),
             ("Sub", 4, 14, neg(3, 4), 1), ("Mul", 4, 5, 3, 15), ("Mul", 4, 5, 10, 2), ("Mul", 4, neg(1, 4), neg(5, 4), 5),
             ("Mul", 127, (1 << 127) - 1, 3, (1 << 127) - 3), ("UnsignedDiv", 4, 5, 3, 1), ("UnsignedDiv", 4, 5, 10, 0),
             ("SignedDiv", 32, 5, neg(10, 32), 0), ("SignedDiv", 32, 5, neg(1, 32), neg(5, 32)), ("SignedDiv", 32, neg(5, 32), neg(1, 32), 5)]
    for op, bits, a, b, want in cases:
        circ = Circuit(3, [Brillig(inputs=[W(1), W(2)], outputs=[3], bytecode=[("BinaryIntOp", 0, op, bits, 0, 1), ("Stop",)])])
        batch = acvm_amd.Batch(acvm_amd.Circuit(circ.to_bytes()), 1, [1, 2])
        batch.set_initial_witness(values_from_rows([[a, b]]))
        assert batch.solve() == 0
        vals, asg = batch.witness(3)
        assert asg[0] == 1 and int.from_bytes(vals[0].tobytes(), "big") == want, (op, bits, a, b)


def test_control_flow_divergent_loop_and_calls(oracle):
    # r0 = n (input). sum = 0; i = 0; while i != n: i += 1; sum += i*i  (per-lane trip count) ; then call a subroutine twice
    bc = [("Const", 1, 0), ("Const", 2, 0), ("Const", 3, 1),          # 0-2: sum, i, one
          ("BinaryFieldOp", 4, "Equals", 2, 0),                        # 3: i == n ?
          ("JumpIf", 4, 9),                                            # 4
          ("BinaryFieldOp", 2, "Add", 2, 3),                           # 5: i += 1
          ("BinaryFieldOp", 5, "Mul", 2, 2),                           # 6
          ("BinaryFieldOp", 1, "Add", 1, 5),                           # 7: sum += i*i
          ("Jump", 3),                                                 # 8
          ("Call", 13), ("Call", 13),                                  # 9, 10
          ("Mov", 0, 1), ("Stop",),                                    # 11, 12
          ("BinaryFieldOp", 1, "Add", 1, 1), ("Return",)]              # 13, 14: sum *= 2
    circ = Circuit(2, [Brillig(inputs=[W(1)], outputs=[2], bytecode=bc)])
    rows = [[n] for n in list(range(0, 40)) + [100, 257, 1000]]
    both_paths(oracle, circ, [1], rows)


def test_trap_return_and_jump_out_of_range(oracle):
    trap = Brillig(inputs=[W(1), W(2)], outputs=[3], bytecode=[("BinaryFieldOp", 2, "Equals", 0, 1), ("JumpIf", 2, 5), ("Call", 4), ("Stop",), ("Trap",), ("Stop",)])
    ret = Brillig(inputs=[W(1)], outputs=[4], bytecode=[("JumpIfNot", 0, 2), ("Return",), ("Stop",)])
    far = Brillig(inputs=[W(2)], outputs=[5], bytecode=[("JumpIf", 0, 77), ("Const", 0, 9)])  # jump past the end = finished
    circ = Circuit(5, [trap, ret, far])
    rows = [[1, 1], [1, 2], [0, 0], [3, 3], [0, 5]]
    ores, _ = both_paths(oracle, circ, [1, 2], rows)
    assert ores[1].err == oracle.E_BRILLIG_FAILED and list(ores[1].call_stack[:ores[1].n_call_stack]) == [2, 4]
    assert ores[0].err == oracle.E_BRILLIG_FAILED and ores[0].opcode_index == 1  # Return with an empty call stack


def test_memory_arrays_and_predicate(oracle):
    r = random.Random(9)
    # inputs: array of 4 + scalar k; reverse the array in memory behind it, write k on top, output the 5 cells
    bc = [("Const", 2, 4), ("Const", 3, 1), ("Const", 4, 0), ("Const", 5, 8),        # n, one, i, dst base
          ("BinaryFieldOp", 6, "Equals", 4, 2), ("JumpIf", 6, 14),                    # 4, 5
          ("BinaryFieldOp", 7, "Add", 0, 4), ("Load", 8, 7),                           # 6, 7: v = mem[ptr + i]
          ("BinaryFieldOp", 9, "Sub", 2, 4), ("BinaryFieldOp", 9, "Sub", 9, 3), ("BinaryFieldOp", 9, "Add", 9, 5),  # 8-10: dst + n-1-i
          ("Store", 9, 8), ("BinaryFieldOp", 4, "Add", 4, 3), ("Jump", 4),             # 11-13
          ("BinaryFieldOp", 9, "Add", 5, 2), ("Store", 9, 1), ("Mov", 0, 5), ("Stop",)]  # 14-17
    br = Brillig(inputs=[[W(1), W(2), E([(1, 1, 2)], [(2, 3)], 5), W(4)], W(5)], outputs=[[10, 11, 12, 13, 14]], bytecode=bc, predicate=W(6))
    oob = Brillig(inputs=[W(5)], outputs=[20], bytecode=[("Load", 0, 0), ("Stop",)])   # reads mem[k] of an empty memory
    circ = Circuit(20, [br, oob])
    rows = [[r.randrange(P) for _ in range(5)] + [r.randrange(2)] for _ in range(66)]
    both_paths(oracle, circ, [1, 2, 3, 4, 5, 6], rows)


def test_unknown_input_and_predicate(oracle):
    a = Brillig(inputs=[W(1), W(9)], outputs=[3], bytecode=[("Stop",)])               # w9 unassigned -> TooManyUnknowns
    b = Brillig(inputs=[W(1)], outputs=[4], bytecode=[("Stop",)], predicate=W(8))      # predicate unassigned -> MissingAssignment
    ores, _ = both_paths(oracle, Circuit(9, [a]), [1], [[1], [2]])
    assert ores[0].err == oracle.E_TOO_MANY_UNKNOWNS
    ores, _ = both_paths(oracle, Circuit(9, [b]), [1], [[1], [2]])
    assert ores[0].err == oracle.E_MISSING_ASSIGNMENT and ores[0].aux0 == 8


def test_output_conflict(oracle):
    br = Brillig(inputs=[W(1)], outputs=[2], bytecode=[("BinaryFieldOp", 0, "Add", 0, 0), ("Stop",)])
    ores, _ = both_paths(oracle, Circuit(2, [br]), [1, 2], [[3, 6], [3, 7]])
    assert ores[0].status == 0 and ores[1].err == oracle.E_UNSATISFIED


@pytest.mark.parametrize("name,n", [("Sha256", 11), ("Sha256", 70), ("Blake2s", 5), ("Keccak256", 140), ("HashToField128Security", 33)])
def test_hash_black_box_ops(oracle, name, n):
    r = random.Random(n)
    ids = list(range(1, n + 1))
    if name == "HashToField128Security":
        bc = [("Const", 1, n), ("BlackBox", name, 0, 1, 2), ("Mov", 0, 2), ("Stop",)]
        outs = [n + 1]
    else:
        bc = [("Const", 1, n), ("Const", 2, 200), ("BlackBox", name, 0, 1, 2, 32), ("Mov", 0, 2), ("Stop",)]
        outs = [list(range(n + 1, n + 33))]
    circ = Circuit(n + 40, [Brillig(inputs=[[W(w) for w in ids]], outputs=outs, bytecode=bc)])
    rows = [[r.randrange(256) for _ in range(n)] for _ in range(40)]
    rows[1] = [r.randrange(P) for _ in range(n)]  # only the last byte of each cell is hashed
    both_paths(oracle, circ, ids, rows)


def test_grumpkin_black_box_ops(oracle):
    import ctypes as C
    r = random.Random(77)
    fixed = Brillig(inputs=[W(1), W(2)], outputs=[[10, 11]], bytecode=[("Const", 2, 50), ("BlackBox", "FixedBaseScalarMul", 0, 1, 2, 2), ("Mov", 0, 2), ("Stop",)])
    ped = Brillig(inputs=[[W(3), W(4)], W(5)], outputs=[[12, 13]],
                  bytecode=[("Const", 2, 2), ("Const", 3, 60), ("BlackBox", "Pedersen", 0, 2, 1, 3, 2), ("Mov", 0, 3), ("Stop",)])
    circ = Circuit(13, [ped, fixed])
    rows = [[r.randrange(1 << 128), r.randrange(1 << 125), r.randrange(P), r.randrange(P), 0] for _ in range(36)]
    rows[0][0] = 1 << 128          # Limb ... is not less than 2^128 -> BrilligFunctionFailed
    rows[1][1] = 1 << 130
    rows[2][4] = 3                 # hash_index 3
    rows[3][4] = 1 << 32           # domain separator does not fit u32
    rows[4][0:2] = [(1 << 128) - 1, (1 << 128) - 1]  # not a valid grumpkin scalar
    ores, _ = run_both(oracle, circ, [1, 2, 3, 4, 5], rows)
    run_both(oracle, circ, [1, 2, 3, 4, 5], rows[:8], force_slow=True)
    assert ores[0].err == oracle.E_BRILLIG_FAILED and ores[3].err == oracle.E_BRILLIG_FAILED and ores[4].err == oracle.E_BRILLIG_FAILED
    # Schnorr through Brillig: pk, message and signature arrays in memory
    msg = bytes(range(10))
    out = C.create_string_buffer(128)
    assert oracle.lib().oracle_schnorr_sign((12345).to_bytes(32, "big"), (6789).to_bytes(32, "big"), msg, len(msg), out) == 0
    pkx, pky, sig = int.from_bytes(out.raw[:32], "big"), int.from_bytes(out.raw[32:64], "big"), list(out.raw[64:128])
    ids = list(range(1, 2 + 10 + 64 + 1))
    sch = Brillig(inputs=[W(1), W(2), [W(w) for w in ids[2:12]], [W(w) for w in ids[12:76]]], outputs=[90],
                  bytecode=[("Const", 4, 10), ("Const", 5, 64), ("BlackBox", "SchnorrVerify", 0, 1, 2, 4, 3, 5, 6), ("Mov", 0, 6), ("Stop",)])
    good = [pkx, pky] + list(msg) + sig
    bad = list(good)
    bad[20] ^= 1
    ores, _ = run_both(oracle, Circuit(90, [sch]), ids, [good, bad, good, bad])
    run_both(oracle, Circuit(90, [sch]), ids, [good, bad], force_slow=True)


@pytest.mark.parametrize("curve", [0, 1])
def test_ecdsa_black_box_ops(oracle, curve):
    """BlackBoxOp::EcdsaSecp256k1 / r1 (brillig_vm/src/black_box.rs:74-131): digest vector + three arrays in VM memory."""
    from ecdsa_ref import CURVES, public_key, sign
    r = random.Random(60 + curve)
    c = CURVES[curve]
    be = lambda v: list(int(v).to_bytes(32, "big"))  # noqa: E731
    name = "EcdsaSecp256k1" if curve == 0 else "EcdsaSecp256r1"
    ids = list(range(1, 161))  # msg 32 | pkx 32 | pky 32 | sig 64
    inputs = [[W(w) for w in ids[:32]], [W(w) for w in ids[32:64]], [W(w) for w in ids[64:96]], [W(w) for w in ids[96:160]]]
    # registers after the inputs: r0..r3 = array pointers; r4 = 32 (message length)
    bc = [("Const", 4, 32), ("BlackBox", name, 0, 4, 1, 32, 2, 32, 3, 64, 5), ("Mov", 0, 5), ("Stop",)]
    circ = Circuit(170, [Brillig(inputs=inputs, outputs=[165], bytecode=bc)])
    rows = []
    for i in range(6):
        sk, k, z = r.randrange(1, c["n"]), r.randrange(1, c["n"]), r.randrange(c["n"])
        Q = public_key(curve, sk)
        rr, ss = sign(curve, sk, k, z)
        rows.append(be(z) + be(Q[0]) + be(Q[1]) + be(rr) + be(ss))
        rows.append(be(z ^ 2) + be(Q[0]) + be(Q[1]) + be(rr) + be(ss))
    rows.append(be(5) + be(c["p"]) + be(1) + be(1) + be(1))  # x >= p: the reference panics
    ores, _ = both_paths(oracle, circ, ids, rows)
    assert ores[0].status == 0 and ores[12].err == oracle.E_PANIC
    # wrong array size -> BlackBoxResolutionError::Failed -> BrilligFunctionFailed
    bad = [("Const", 4, 32), ("BlackBox", name, 0, 4, 1, 31, 2, 32, 3, 64, 5), ("Mov", 0, 5), ("Stop",)]
    ores, _ = both_paths(oracle, Circuit(170, [Brillig(inputs=inputs, outputs=[165], bytecode=bad)]), ids, rows[:2])
    assert ores[0].err == oracle.E_BRILLIG_FAILED
