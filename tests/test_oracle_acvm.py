"""Pins oracle/pwg.c + brillig_vm.c against the behaviours the reference's own tests assert:
acvm/src/pwg/arithmetic.rs:242-281, memory_op.rs:138-258, directives/mod.rs:134-149, acvm/tests/solver.rs,
brillig_vm/src/arithmetic.rs:149-234, brillig_vm/src/black_box.rs:176-210, acvm_js/test/shared/*.ts."""
import pytest

from acvm_amd.acir import (Arithmetic, BlackBoxFuncCall, Brillig, Circuit, Expression, FunctionInput, MemoryInit,
                           MemoryOp, P, QuotientDirective, ToLeRadix)

M1 = P - 1


def run(oracle, circ, iw, backend=1):
    c = oracle.Circuit(circ.to_bincode())
    a = oracle.ACVM(c, iw, backend)
    st = a.solve()
    return a, st


def test_arithmetic_smoke(oracle):
    # arithmetic.rs:242-281: a = b + c + d ; e = a + b with (2,1,1) -> a = 4, e = 6
    circ = Circuit(4, [Expression([], [(1, 0), (M1, 1), (M1, 2), (M1, 3)], 0), Expression([], [(1, 4), (M1, 0), (M1, 1)], 0)])
    a, st = run(oracle, circ, {1: 2, 2: 1, 3: 1})
    assert st == oracle.ST_SOLVED
    wm = a.witness_map()
    assert wm[0] == 4 and wm[4] == 6


def test_empty_circuit_is_solved_at_construction(oracle):
    a, st = run(oracle, Circuit(1, []), {1: 5})
    assert st == oracle.ST_SOLVED and a.witness_map() == {1: 5}


def test_unsatisfied_opcode_resolved(oracle):
    # solver.rs:490-525: a = b + c + d with (4,2,1,2) -> UnsatisfiedConstrain{Resolved(Acir(0))}
    circ = Circuit(3, [Expression([], [(1, 0), (M1, 1), (M1, 2), (M1, 3)], 0)])
    a, st = run(oracle, circ, {0: 4, 1: 2, 2: 1, 3: 2})
    assert st == oracle.ST_FAILURE
    assert a.result().as_tuple() == (oracle.ST_FAILURE, oracle.E_UNSATISFIED, 0, 0, 0)


def test_too_many_unknowns_and_zero_coefficient_drop(oracle):
    # two unknown linear terms -> ExpressionHasTooManyUnknowns (arithmetic.rs:38-42)
    a, st = run(oracle, Circuit(3, [Expression([], [(1, 1), (1, 2), (1, 3)], 0)]), {1: 1})
    assert a.result().as_tuple()[:3] == (oracle.ST_FAILURE, oracle.E_TOO_MANY_UNKNOWNS, 0)
    # folded term with zero known multiplicand is dropped (arithmetic.rs:217-221): 5*w1*w2 + w3 = 0, w1 = 0, w3 = 0
    a, st = run(oracle, Circuit(3, [Expression([(5, 1, 2)], [(1, 3)], 0), Expression([], [(1, 2)], 0)]), {1: 0, 3: 0})
    # first opcode is satisfied without assigning w2; second opcode then solves w2 = 0
    assert st == oracle.ST_SOLVED and a.witness_map()[2] == 0
    # same but constant does not vanish -> Unsatisfied at opcode 0
    a, st = run(oracle, Circuit(3, [Expression([(5, 1, 2)], [(1, 3)], 7)]), {1: 0, 3: 0})
    assert a.result().as_tuple()[:3] == (oracle.ST_FAILURE, oracle.E_UNSATISFIED, 0)
    # the same unknown twice counts as two unknowns (arithmetic.rs:188-201)
    a, st = run(oracle, Circuit(3, [Expression([(1, 1, 2)], [(1, 2)], 0)]), {1: 3})
    assert a.result().as_tuple()[:3] == (oracle.ST_FAILURE, oracle.E_TOO_MANY_UNKNOWNS, 0)
    # two residual mul terms: the reference panics (arithmetic.rs:142)
    a, st = run(oracle, Circuit(4, [Expression([(1, 1, 2), (1, 3, 4)], [], 0)]), {})
    assert a.result().err == oracle.E_PANIC


def test_memory_operations(oracle):
    # solver.rs:610-648
    circ = Circuit(8, [MemoryInit(0, [1, 2, 3, 4, 5]),
                       MemoryOp(0, Expression.constant(0), Expression.from_witness(6), Expression.from_witness(7)),
                       Expression([], [(1, 7), (M1, 8)], 1)])
    a, st = run(oracle, circ, {1: 1, 2: 2, 3: 3, 4: 4, 5: 5, 6: 4})
    assert st == oracle.ST_SOLVED and a.witness_map()[8] == 6


def _mem_trace(ops, pred=None):
    return [MemoryInit(0, [1, 2])] + [MemoryOp(0, *o, predicate=pred) for o in ops]


def test_memory_op_unit_tests(oracle):
    W = Expression.from_witness
    K = Expression.constant
    # memory_op.rs:138-170 write then read
    circ = Circuit(4, _mem_trace([(K(1), K(1), W(3)), (K(0), K(1), W(4))]))
    a, st = run(oracle, circ, {1: 1, 2: 1, 3: 2})
    assert st == oracle.ST_SOLVED and a.witness_map()[4] == 2
    # :172-206 out of bounds: index 2 on array of size 2
    circ = Circuit(4, _mem_trace([(K(1), K(1), W(3)), (K(1), K(2), W(3)), (K(0), K(2), W(4))]))
    a, st = run(oracle, circ, {1: 1, 2: 1, 3: 2})
    assert a.result().as_tuple() == (oracle.ST_FAILURE, oracle.E_INDEX_OOB, 2, 2, 2)
    # :208-233 zero predicate on a read: value 0, no bounds check
    circ = Circuit(4, _mem_trace([(K(1), K(1), W(3)), (K(0), K(2), W(4))], pred=K(0)))
    a, st = run(oracle, circ, {1: 1, 2: 1, 3: 2})
    assert st == oracle.ST_SOLVED and a.witness_map()[4] == 0
    # :235-258 zero predicate on a write: skipped, later read sees the initial value
    ops = [MemoryInit(0, [1, 2]), MemoryOp(0, K(1), K(1), W(3), predicate=K(0)), MemoryOp(0, K(0), K(1), W(4))]
    a, st = run(oracle, Circuit(4, ops), {1: 1, 2: 1, 3: 2})
    assert st == oracle.ST_SOLVED and a.witness_map()[4] == 1


def test_quotient_and_radix_directives(oracle):
    W = Expression.from_witness
    # directives/mod.rs:134-149: 0 / 0 with predicate 1 -> q = r = 0
    a, st = run(oracle, Circuit(4, [QuotientDirective(W(1), W(2), 3, 4, Expression.constant(1))]), {1: 0, 2: 0})
    assert st == oracle.ST_SOLVED and a.witness_map()[3] == 0 and a.witness_map()[4] == 0
    a, st = run(oracle, Circuit(4, [QuotientDirective(W(1), W(2), 3, 4)]), {1: P - 1, 2: 12345})
    assert (a.witness_map()[3], a.witness_map()[4]) == divmod(P - 1, 12345)
    a, st = run(oracle, Circuit(4, [QuotientDirective(W(1), W(2), 3, 4, Expression.constant(0))]), {1: 99, 2: 7})
    assert (a.witness_map()[3], a.witness_map()[4]) == (0, 0)
    # ToLeRadix: digits little-endian, zero padded; too many digits -> Unsatisfied (directives/mod.rs:60-87)
    a, st = run(oracle, Circuit(6, [ToLeRadix(W(1), [2, 3, 4, 5], 256)]), {1: 0x01020304})
    assert [a.witness_map()[i] for i in (2, 3, 4, 5)] == [4, 3, 2, 1]
    a, st = run(oracle, Circuit(6, [ToLeRadix(W(1), [2, 3], 2)]), {1: 2})
    assert [a.witness_map()[i] for i in (2, 3)] == [0, 1]
    a, st = run(oracle, Circuit(6, [ToLeRadix(W(1), [2, 3], 2)]), {1: 4})
    assert a.result().as_tuple()[:3] == (oracle.ST_FAILURE, oracle.E_UNSATISFIED, 0)
    a, st = run(oracle, Circuit(6, [ToLeRadix(W(1), [2, 3, 4], 10)]), {1: 0})
    assert [a.witness_map()[i] for i in (2, 3, 4)] == [0, 0, 0]


def test_range_and_logic(oracle):
    FI = FunctionInput
    a, st = run(oracle, Circuit(3, [BlackBoxFuncCall("RANGE", dict(input=FI(1, 8)))]), {1: 255})
    assert st == oracle.ST_SOLVED
    a, st = run(oracle, Circuit(3, [BlackBoxFuncCall("RANGE", dict(input=FI(1, 8)))]), {1: 256})
    assert a.result().as_tuple()[:3] == (oracle.ST_FAILURE, oracle.E_UNSATISFIED, 0)
    a, st = run(oracle, Circuit(3, [BlackBoxFuncCall("RANGE", dict(input=FI(1, 8)))]), {})
    assert a.result().as_tuple() == (oracle.ST_FAILURE, oracle.E_MISSING_ASSIGNMENT, 0, 1, 0)
    circ = Circuit(4, [BlackBoxFuncCall("AND", dict(lhs=FI(1, 16), rhs=FI(2, 16), output=3)),
                       BlackBoxFuncCall("XOR", dict(lhs=FI(1, 16), rhs=FI(2, 16), output=4))])
    a, st = run(oracle, circ, {1: 0xF0F0F, 2: 0x3C3C})
    assert a.witness_map()[3] == (0x0F0F & 0x3C3C) and a.witness_map()[4] == (0x0F0F ^ 0x3C3C)


def test_hash_opcodes(oracle):
    import hashlib
    FI = FunctionInput
    msg = b"hello world"
    ins = [FI(i + 1, 8) for i in range(len(msg))]
    outs = list(range(20, 52))
    iw = {i + 1: b for i, b in enumerate(msg)}
    a, st = run(oracle, Circuit(60, [BlackBoxFuncCall("SHA256", dict(inputs=ins, outputs=outs))]), iw)
    assert bytes(a.witness_map()[w] for w in outs) == hashlib.sha256(msg).digest()
    a, st = run(oracle, Circuit(60, [BlackBoxFuncCall("Blake2s", dict(inputs=ins, outputs=outs))]), iw)
    assert bytes(a.witness_map()[w] for w in outs) == hashlib.blake2s(msg).digest()
    a, st = run(oracle, Circuit(60, [BlackBoxFuncCall("HashToField128Security", dict(inputs=ins, output=20))]), iw)
    assert a.witness_map()[20] == int.from_bytes(hashlib.blake2s(msg).digest(), "big") % P
    # a 32-bit input contributes 4 bytes little-endian (hash.rs:63, generic_ark.rs:305-317)
    a, st = run(oracle, Circuit(60, [BlackBoxFuncCall("SHA256", dict(inputs=[FI(1, 32)], outputs=outs))]), {1: 0x01020304})
    assert bytes(a.witness_map()[w] for w in outs) == hashlib.sha256(bytes([4, 3, 2, 1])).digest()
    # wrong output count (hash.rs:39-44)
    a, st = run(oracle, Circuit(60, [BlackBoxFuncCall("SHA256", dict(inputs=ins, outputs=outs[:31]))]), iw)
    r = a.result()
    assert (r.err, r.aux0, r.message) == (oracle.E_BLACKBOX_FAILED, 3, b"Expected 32 outputs but encountered 31")
    # variable-length keccak takes more bytes than available (hash.rs:68-80)
    a, st = run(oracle, Circuit(60, [BlackBoxFuncCall("Keccak256VariableLength",
                                                      dict(inputs=ins, var_message_size=FI(15, 32), outputs=outs))]), {**iw, 15: 12})
    r = a.result()
    assert r.err == oracle.E_BLACKBOX_FAILED and r.aux0 == 11 and b"12 > 11" in r.message


def test_stubbed_backend_panics_when_hit(oracle):
    FI = FunctionInput
    circ = Circuit(4, [BlackBoxFuncCall("Pedersen", dict(inputs=[FI(1, 254)], domain_separator=0, outputs=(2, 3)))])
    a, st = run(oracle, circ, {1: 1}, backend=oracle.BACKEND_STUBBED)
    assert a.result().err == oracle.E_PANIC
    a, st = run(oracle, circ, {1: 1}, backend=oracle.BACKEND_DUMMY)
    assert st == oracle.ST_SOLVED and (a.witness_map()[2], a.witness_map()[3]) == (2, 3)


def test_oracle_dependent_execution(oracle):
    # solver.rs:308-426
    br = Brillig(inputs=[Expression.from_witness(1), Expression(), Expression.from_witness(2)], outputs=[1, 4, 2, 4],
                 bytecode=[("ForeignCall", "invert", [("Register", 1)], [("Register", 0)]),
                           ("ForeignCall", "invert", [("Register", 3)], [("Register", 2)])])
    circ = Circuit(4, [Expression([], [(M1, 1), (1, 2)], 0), br, Expression([], [(M1, 3), (1, 4)], 0)])
    c = oracle.Circuit(circ.to_bincode())
    a = oracle.ACVM(c, {1: 2, 2: 2}, oracle.BACKEND_STUBBED)
    for _ in range(2):
        assert a.solve() == oracle.ST_REQUIRES_FOREIGN_CALL
        assert a.instruction_pointer() == 1
        fn, inputs = a.get_pending_foreign_call()
        assert fn == "invert" and len(inputs) == 1
        a.resolve_pending_foreign_call([pow(inputs[0][0], P - 2, P)])
    # w3 is never assigned by the brillig outputs (w_y_inv twice), so the last opcode solves w3 = w4
    assert a.solve() == oracle.ST_SOLVED
    with pytest.raises(RuntimeError):
        a.resolve_pending_foreign_call([1])  # reference panics: not waiting


def test_brillig_zero_predicate_zeroes_outputs(oracle):
    # solver.rs:428-489 / brillig.rs:34-37
    br = Brillig(inputs=[Expression.from_witness(1)], outputs=[2, [3, 4]],
                 bytecode=[("ForeignCall", "invert", [("Register", 1)], [("Register", 0)])], predicate=Expression.constant(0))
    a, st = run(oracle, Circuit(4, [br]), {1: 7})
    assert st == oracle.ST_SOLVED and [a.witness_map()[i] for i in (2, 3, 4)] == [0, 0, 0]


def test_brillig_trap_call_stack(oracle):
    # solver.rs:527-608
    br = Brillig(inputs=[Expression.from_witness(4), Expression.from_witness(5)], outputs=[6],
                 bytecode=[("BinaryFieldOp", 2, "Equals", 0, 1), ("JumpIf", 2, 3), ("Trap",), ("Stop",)],
                 predicate=Expression.constant(1))
    circ = Circuit(6, [br, Expression([], [(1, 0), (M1, 1), (M1, 2), (M1, 3)], 0)])
    a, st = run(oracle, circ, {0: 4, 1: 2, 2: 1, 3: 2, 4: 0, 5: 1, 6: 0})
    r = a.result()
    assert st == oracle.ST_FAILURE and r.err == oracle.E_BRILLIG_FAILED and r.opcode_index == 0
    assert r.message == b"explicit trap hit in brillig" and list(r.call_stack[:r.n_call_stack]) == [2]


def _int_op(oracle, op, bits, a, b):
    br = Brillig(inputs=[Expression.from_witness(1), Expression.from_witness(2)], outputs=[3],
                 bytecode=[("BinaryIntOp", 0, op, bits, 0, 1), ("Stop",)])
    acvm, st = run(oracle, Circuit(3, [br]), {1: a, 2: b})
    return acvm.witness_map().get(3) if st == oracle.ST_SOLVED else acvm.result()


def test_brillig_int_ops_known_answers(oracle):
    # brillig_vm/src/arithmetic.rs:149-234
    neg = lambda x, bits: (1 << bits) - x  # noqa: E731
    for a, b, r in [(5, 10, 15), (10, 10, 4), (5, neg(3, 4), 2), (neg(3, 4), 1, neg(2, 4)), (5, neg(6, 4), neg(1, 4))]:
        assert _int_op(oracle, "Add", 4, a, b) == r
    for a, b, r in [(5, 3, 2), (5, 10, neg(5, 4)), (5, neg(3, 4), 8), (neg(3, 4), 2, neg(5, 4)), (14, neg(3, 4), 1)]:
        assert _int_op(oracle, "Sub", 4, a, b) == r
    for a, b, r in [(5, 3, 15), (5, 10, 2), (neg(1, 4), neg(5, 4), 5), (neg(1, 4), 5, neg(5, 4)), (neg(2, 4), 7, neg(14, 4))]:
        assert _int_op(oracle, "Mul", 4, a, b) == r
    a127 = (1 << 127) - 1
    assert _int_op(oracle, "Mul", 127, a127, 3) == a127 - 2
    assert _int_op(oracle, "UnsignedDiv", 4, 5, 3) == 1 and _int_op(oracle, "UnsignedDiv", 4, 5, 10) == 0
    for a, b, r in [(5, neg(10, 32), 0), (5, neg(1, 32), neg(5, 32)), (neg(5, 32), neg(1, 32), 5)]:
        assert _int_op(oracle, "SignedDiv", 32, a, b) == r
    # comparisons, bit ops, shifts
    assert _int_op(oracle, "LessThan", 8, 3, 300) == 1  # operands reduced mod 2^8 first: 3 < 44
    assert _int_op(oracle, "LessThan", 8, 3, 258) == 0  # 3 < 2 is false
    assert _int_op(oracle, "Shl", 8, 0x81, 1) == 0x02 and _int_op(oracle, "Shr", 8, 0x181, 1) == 0xC0
    assert _int_op(oracle, "Xor", 16, 0xFFFF0, 0x0FF0F) == (0xFFFF0 ^ 0x0FF0F) & 0xFFFF
    assert _int_op(oracle, "UnsignedDiv", 8, 5, 0).err == oracle.E_PANIC  # division by zero panics (num-bigint)


def test_brillig_sha256_black_box(oracle):
    # brillig_vm/src/black_box.rs:176-210: sha256("hello world") through the Brillig BlackBox op
    msg = b"hello world"
    br = Brillig(inputs=[[Expression.constant(b) for b in msg]], outputs=[[10 + i for i in range(32)]],
                 bytecode=[("Const", 1, len(msg)), ("Const", 2, 100), ("BlackBox", "Sha256", 0, 1, 2, 32), ("Mov", 0, 2), ("Stop",)])
    a, st = run(oracle, Circuit(50, [br]), {})
    assert st == oracle.ST_SOLVED
    assert bytes(a.witness_map()[10 + i] for i in range(32)).hex() == "b94d27b9934d3e08a52e52d7da7dabfac484efe37a5380ee9088f7ace2efcde9"


def test_acvm_js_fixtures(oracle, golden):
    """bytecode + initial witness -> expected witness map for every acvm_js fixture, incl. foreign-call round trips."""
    for name, fx in golden["acvm_js"].items():
        if "bytecode" not in fx:
            continue
        c = oracle.Circuit(bytes(fx["bytecode"]))
        a = oracle.ACVM(c, {int(k): int(v, 16) for k, v in fx["initialWitnessMap"].items()})
        st = a.solve()
        while st == oracle.ST_REQUIRES_FOREIGN_CALL:
            fn, inputs = a.get_pending_foreign_call()
            assert fn == fx["oracleCallName"]
            assert inputs == [[int(x, 16) for x in grp] for grp in fx["oracleCallInputs"]]
            a.resolve_pending_foreign_call([int(x, 16) if isinstance(x, str) else [int(y, 16) for y in x] for x in fx["oracleResponse"]])
            st = a.solve()
        assert st == oracle.ST_SOLVED, name
        wm = a.witness_map()
        if "expectedWitnessMap" in fx:
            assert wm == {int(k): int(v, 16) for k, v in fx["expectedWitnessMap"].items()}, name
        else:
            assert wm[fx["resultWitness"]] == int(fx["expectedResult"], 16)


def test_config1_fixture(oracle):
    """BASELINE config 1 (1 000 gates, one instance): the oracle against tests/golden/config1.json, which was produced by an
    independent Python big-integer solve (tests/golden/make_config1_fixture.py)."""
    import hashlib
    import json
    import os
    from acvm_amd import synth
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config1.json")))
    circ, ids = synth.arithmetic_circuit(fx["gates"], seed=fx["seed"])
    res, asg, vals = oracle.solve_batch(oracle.Circuit(circ.to_bytes()), ids, bytes.fromhex(fx["inputs_be32_hex"]), 1)
    assert res[0].status == 0 and asg[0, 1:fx["n_witnesses"] + 1].all()
    assert hashlib.sha256(vals[0, 1:fx["n_witnesses"] + 1].tobytes()).hexdigest() == fx["sha256_of_witnesses_1_to_n"]
    for w, v in fx["witnesses"].items():
        assert vals[0, int(w)].tobytes().hex() == v


def ref_int_op(op, a, b, bits):
    """evaluate_binary_bigint_op (brillig_vm/src/arithmetic.rs:23-98) with Python integers, then FieldElement::from_be_bytes_reduce: the
    reference's BigUint arithmetic takes ANY bit_size. Returns the field value, or None where the reference panics."""
    m = 1 << bits

    def signed(x):
        return x if x < (1 << (bits - 1)) else x - 2 * (1 << (bits - 1))
    if op == "Add":
        r = (a + b) % m
    elif op == "Sub":
        r = (m + a - b) % m if m + a - b >= 0 else None
    elif op == "Mul":
        r = (a * b) % m
    elif op == "UnsignedDiv":
        r = (a % m) // (b % m) if b % m else None
    elif op == "SignedDiv":
        if bits == 0 or signed(b) == 0:
            return None
        sa, sb = signed(a), signed(b)
        q = abs(sa) // abs(sb) * (1 if (sa < 0) == (sb < 0) else -1)  # BigInt division truncates toward zero
        r = q if q >= 0 else (m + q if m + q >= 0 else None)
    elif op in ("Equals", "LessThan", "LessThanEquals"):
        x, y = a % m, b % m
        r = int(x == y if op == "Equals" else x < y if op == "LessThan" else x <= y)
    elif op in ("And", "Or", "Xor"):
        r = (a & b if op == "And" else a | b if op == "Or" else a ^ b) % m
    else:
        return None
    return None if r is None else r % P


@pytest.mark.parametrize("bits", [255, 256, 257, 300, 400, 507, 508, 509, 512, 1000, (1 << 20) + 3])
def test_brillig_int_ops_at_any_bit_size(oracle, bits):
    """bit sizes beyond 256: the reference's BigUint has no width limit (arithmetic.rs:23-34); checked against Python integers"""
    import random
    r = random.Random(bits)
    vals = [0, 1, 2, P - 1, P - 2, (1 << 253) + 12345, (1 << 128) - 1, 1 << 200] + [r.randrange(P) for _ in range(6)]
    for op in ("Add", "Sub", "Mul", "UnsignedDiv", "SignedDiv", "Equals", "LessThan", "LessThanEquals", "And", "Or", "Xor"):
        br = Brillig(inputs=[Expression.from_witness(1), Expression.from_witness(2)], outputs=[3], bytecode=[("BinaryIntOp", 0, op, bits, 0, 1), ("Stop",)])
        oc = oracle.Circuit(Circuit(3, [br]).to_bytes())
        for a in vals:
            for b in vals[:9]:
                acvm = oracle.ACVM(oc, {1: a, 2: b})
                st = acvm.solve()
                want = ref_int_op(op, a, b, bits)
                if want is None:
                    assert st == oracle.ST_FAILURE, (op, bits, a, b)
                else:
                    assert st == oracle.ST_SOLVED and acvm.witness_map()[3] == want, (op, bits, a, b, acvm.witness_map().get(3), want)
