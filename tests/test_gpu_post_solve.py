"""What a caller does right after solve, on solved batches: the error string of acvm_js/src/execute.rs:79-108
(assert message for the failing location, else the Display text of OpcodeResolutionError, acvm/src/pwg/mod.rs:100-114) and
extract_indices / getReturnWitness / getPublicWitness (acvm_js/src/public_witness.rs)."""
import numpy as np
import pytest

import acvm_amd
from acvm_amd.acir import P, BlackBoxFuncCall as BB, Circuit, Expression as E, FunctionInput as FI, MemoryInit, MemoryOp
from acvm_amd.synth import be32

pytestmark = pytest.mark.gpu


def solve(circ, ids, rows):
    c = acvm_amd.Circuit(circ.to_bytes())
    b = acvm_amd.Batch(c, len(rows), ids)
    b.set_initial_witness(b"".join(be32(v) for row in rows for v in row))
    b.solve()
    return c, b


def test_error_strings_and_assert_messages():
    # opcode 0: w3 = w1 * w2; opcode 1: assert w3 == w4 (message); opcode 2: RANGE(w1, 8) (no message); opcode 3: memory read
    ops = [E([(1, 1, 2)], [(P - 1, 3)], 0), E([], [(1, 3), (P - 1, 4)], 0), BB("RANGE", {"input": FI(1, 8)}), MemoryInit(0, [1, 2]),
           MemoryOp(0, E.constant(0), E.from_witness(5), E.from_witness(6))]
    circ = Circuit(6, ops, private_parameters=[1, 2, 4, 5], return_values=[3],
                   assert_messages=[(1, "product check failed"), (4, "index in range")])
    rows = [[3, 5, 15, 1], [3, 5, 16, 0], [300, 1, 300, 0], [3, 5, 15, 2]]
    c, b = solve(circ, [1, 2, 4, 5], rows)
    res = b.results()
    assert [r.status for r in res] == [0, 2, 2, 2]
    assert b.error_string(0) == ""
    assert b.error_string(1) == "Assertion failed: product check failed"
    assert b.error_string(2) == "Cannot satisfy constraint"                       # RANGE at opcode 2 has no message
    assert b.error_string(3) == "Assertion failed: index in range"                # IndexOutOfBounds at opcode 4
    # without the message table the Display text of the error is reported
    plain = Circuit(6, ops, private_parameters=[1, 2, 4, 5], return_values=[3])
    c2, b2 = solve(plain, [1, 2, 4, 5], rows)
    assert b2.error_string(1) == "Cannot satisfy constraint"
    assert b2.error_string(3) == "Index out of bounds, array has size 2, but index was 2"


def test_missing_assignment_and_blackbox_failure_text():
    circ = Circuit(9, [BB("RANGE", {"input": FI(9, 8)})])
    c, b = solve(circ, [1], [[1]])
    assert b.error_string(0) == "Cannot solve opcode: missing assignment for witness index 9"
    circ = Circuit(4, [BB("FixedBaseScalarMul", {"low": FI(1, 128), "high": FI(2, 128), "outputs": [3, 4]})])
    c, b = solve(circ, [1, 2], [[1 << 128, 0]])
    assert b.error_string(0) == ("Failed to solve blackbox function: fixed_base_scalar_mul, reason: Limb "
                                 "0000000000000000000000000000000100000000000000000000000000000000 is not less than 2^128")


def test_extract_return_and_public_witnesses():
    ops = [E([(1, 1, 2)], [(P - 1, 4)], 0), E([], [(1, 4), (1, 3), (P - 1, 5)], 0)]
    circ = Circuit(5, ops, private_parameters=[1, 2], public_parameters=[3], return_values=[5])
    rows = [[2, 3, 4], [5, 6, 7], [P - 1, 2, 1]]
    c, b = solve(circ, [1, 2, 3], rows)
    ret = b.extract(c.witness_set("return_values"))
    assert [int.from_bytes(ret[j, 0].tobytes(), "big") for j in range(3)] == [(r[0] * r[1] + r[2]) % P for r in rows]
    pub = b.extract(c.witness_set("public_inputs"), first=1, n=2)
    assert pub.shape == (2, 2, 32) and int.from_bytes(pub[0, 0].tobytes(), "big") == 7
    asg, full = b.witness_map()
    assert np.array_equal(pub[1, 1], full[2, 5])
    # a failed instance leaves the return value unassigned: extract_indices reports the witness
    bad = Circuit(5, [E([], [(1, 1), (P - 1, 2)], 0)] + ops, private_parameters=[1, 2], public_parameters=[3], return_values=[5])
    c, b = solve(bad, [1, 2, 3], [[2, 2, 4], [2, 3, 4]])
    assert b.results()[1].status == 2
    b.extract([5], first=0, n=1)
    with pytest.raises(acvm_amd.AcvmError, match="Failed to extract witness 4 from witness map. Witness not found."):
        b.extract([4, 5])


def test_solved_witness_map_in_the_wire_format(golden):
    """finalize() + compressWitness: the reference's addition circuit (acvm_js/test/shared/addition.ts) solved on the device,
    its witness map serialised like acir's WitnessMap and read back."""
    fx = golden["acvm_js"]["addition"]
    c = acvm_amd.Circuit(bytes(fx["bytecode"]))
    iw = {int(k): int(v, 16) for k, v in fx["initialWitnessMap"].items()}
    ids = sorted(iw)
    b = acvm_amd.Batch(c, 2, ids)
    b.set_initial_witness(b"".join(be32(iw[i]) for i in ids) * 2)
    assert b.solve() == 0
    got = acvm_amd.decompress_witness(b.witness_map_bytes(1))
    asg, vals = b.witness_map(1, 1)
    assert got == {w: int.from_bytes(vals[0, w].tobytes(), "big") for w in range(asg.shape[1]) if asg[0, w]}
    assert got[fx["resultWitness"]] == int(fx["expectedResult"], 16) and all(got[k] == v for k, v in iw.items())


@pytest.mark.parametrize("n_gates,force_slow", [(3, False), (300, False), (300, True), (700, False)])
def test_witness_map_digest(oracle, n_gates, force_slow):
    """acvm_batch_digest against its definition evaluated with hashlib over the ORACLE's witness map: solved instances (scaled
    columns, the planner's assigned set), failing instances (per-instance assigned set, the map as it stands at the failure),
    every instance through the exact kernels, odd and even numbers of assigned witnesses per segment, several segments."""
    import acvm_amd
    from acvm_amd import synth
    circ, ids = synth.arithmetic_circuit(n_gates, seed=0xD16E57 + n_gates)
    B = 70
    values = synth.witness_batch(B, seed=0xD16E57 + n_gates)   # instances 0..7 are edge cases, some of them fail
    oc = oracle.Circuit(circ.to_bytes())
    ores, oasg, ovals = oracle.solve_batch(oc, ids, values, B)
    batch = acvm_amd.Batch(acvm_amd.Circuit(circ.to_bytes()), B, ids)
    batch.set_force_slow_path(force_slow)
    batch.set_initial_witness(values)
    batch.solve()
    got = batch.digest()
    part = batch.digest(first=5, n=9)
    batch.free()
    assert n_gates < 100 or any(r.status != 0 for r in ores)
    for j in range(B):
        assert bytes(got[j]) == oracle.witness_map_digest(oasg[j], ovals[j]), f"instance {j} (status {ores[j].status})"
    assert np.array_equal(part, got[5:14])


def test_witness_map_digest_with_black_box_outputs(oracle):
    """Digest over a map that holds pinned (hash in / out) and scaled witnesses side by side, and an empty circuit."""
    import acvm_amd
    from acvm_amd import synth
    circ, ids = synth.mixed_circuit(400, seed=0xD16E58)
    B = 66
    values = synth.witness_batch(B, n_in=len(ids), seed=0xD16E58)
    oc = oracle.Circuit(circ.to_bytes())
    ores, oasg, ovals = oracle.solve_batch(oc, ids, values, B)
    batch = acvm_amd.Batch(acvm_amd.Circuit(circ.to_bytes()), B, ids)
    batch.set_initial_witness(values)
    batch.solve()
    got = batch.digest()
    batch.free()
    for j in range(B):
        assert bytes(got[j]) == oracle.witness_map_digest(oasg[j], ovals[j]), f"instance {j} (status {ores[j].status})"


@pytest.mark.parametrize("shape,force_slow", [("arith3", False), ("arith300", False), ("arith300", True), ("arith700", False), ("mixed", False), ("mixed_fold", False)])
def test_witness_map_blake2s_tree_digest(oracle, shape, force_slow):
    """acvm_batch_digest_blake2s (SURVEY 8d's "blake2s over the full witness vector", in tree form) against hashlib over the ORACLE's witness map:
    solved instances (scaled and relaxed columns leave through 1 / scale), failing instances (their own assigned set: the map as it stands), every
    instance through the exact kernels, witness counts that are not multiples of the leaf (an odd last witness, a short last leaf), pinned
    hash outputs beside scaled witnesses; refused with recycled rows."""
    import acvm_amd
    from acvm_amd import synth
    if shape.startswith("arith"):
        n_gates = int(shape[5:])
        circ, ids = synth.arithmetic_circuit(n_gates, seed=0xD16E57 + n_gates)
        values = synth.witness_batch(70, seed=0xD16E57 + n_gates)
    else:
        circ, ids = synth.mixed_circuit(400, seed=0xD16E58)
        values = synth.witness_batch(70, n_in=len(ids), seed=0xD16E58)
    B = 70
    oc = oracle.Circuit(circ.to_bytes())
    ores, oasg, ovals = oracle.solve_batch(oc, ids, values, B)
    gc = acvm_amd.Circuit(circ.to_bytes())
    batch = acvm_amd.Batch(gc, B, ids, fold_digest=shape == "mixed_fold")
    batch.set_force_slow_path(force_slow)
    batch.set_initial_witness(values)
    batch.solve()
    got = batch.digest_blake2s()
    part = batch.digest_blake2s(first=5, n=9)
    fingerprint = batch.digest()
    batch.free()
    for j in range(B):
        assert bytes(got[j]) == oracle.witness_map_blake2s(oasg[j], ovals[j]), f"instance {j} (status {ores[j].status})"
        assert bytes(fingerprint[j]) == oracle.witness_map_digest(oasg[j], ovals[j])
    assert np.array_equal(part, got[5:14])
    if shape == "mixed":
        reuse = acvm_amd.Batch(gc, B, ids, reuse_slots=True, keep=gc.witness_set("return_values"))
        reuse.set_initial_witness(values)
        reuse.solve()
        with pytest.raises(acvm_amd.AcvmError, match="recycled"):
            reuse.digest_blake2s()
        reuse.free()


# ---- OpcodeNotSolvable::ExpressionHasTooManyUnknowns(Expression): the Display text carries the expression (pwg/mod.rs:72-78)
_SUP = "⁰¹²³⁴⁵⁶⁷⁸⁹"


def field_display(v):
    """impl Display for FieldElement, acir_field/src/generic_ark.rs:13-74, restated with Python integers"""
    v %= P
    if v == 0:
        return "0"
    minus = (P - v) % P
    neg = len(str(minus)) < len(str(v))
    s = minus if neg else v
    out = "-" if neg else ""
    if bin(s).count("1") == 1:
        bit = s.bit_length() - 1
        return out + (str(1 << bit) if bit < 4 else "2" + "".join(_SUP[int(d)] for d in str(bit)))
    for power in (64, 32, 16, 8, 4):
        if s % (1 << power) == 0:
            return out + "2" + "".join(_SUP[int(d)] for d in str(power)) + "×" + str(s >> power)
    return out + str(s)


def expr_display(mul, lin, qc):
    """impl Display for Expression (expression/mod.rs:40-48) over Opcode::Arithmetic's Debug text (circuit/opcodes.rs:88-102)"""
    if not mul and len(lin) == 1 and lin[0][0] % P == 1 and qc % P == 0:
        return f"x{lin[0][1]}"
    return "%EXPR [ " + "".join(f"({field_display(c)}, _{a}, _{b}) " for c, a, b in mul) + "".join(f"({field_display(c)}, _{w}) " for c, w in lin) + field_display(qc) + " ]%"


def test_field_display_shapes():
    assert [field_display(v) for v in (0, 1, 2, 8, 16, 1 << 64, P - 1, P - 16, 3 << 64, 5 << 32, 48, 7, 1 << 100)] == \
        ["0", "1", "2", "8", "2⁴", "2⁶⁴", "-1", "-2⁴", "2⁶⁴×3", "2³²×5", "2⁴×3", "7", "2¹⁰⁰"]


def test_too_many_unknowns_quotes_the_evaluated_expression():
    from acvm_amd.acir import Brillig
    # opcode 0: 3*w1*w2 + 5*w1*w8 + 0*w9*w9 + 7*w3 + 2*w8 + 11*w9 - (2^64)*w2 + 100 = 0 with w8, w9 unknown: evaluate() folds the known parts into the
    # constant, turns 5*w1*w8 into a linear term on w8 (dropped where w1 == 0), keeps the unknown linear terms in order
    e0 = E([(3, 1, 2), (5, 1, 8), (0, 9, 9)], [(7, 3), (2, 8), (11, 9), (P - (1 << 64), 2)], 100)
    circ = Circuit(9, [e0])
    rows = [[4, 6, 9], [0, 1, 2], [P - 1, 1 << 70, 5]]
    c, b = solve(circ, [1, 2, 3], rows)
    for j, (w1, w2, w3) in enumerate(rows):
        lin = ([((5 * w1) % P, 8)] if (5 * w1) % P else []) + [(2, 8), (11, 9)]
        qc = (3 * w1 * w2 + 7 * w3 - (1 << 64) * w2 + 100) % P
        assert b.error_string(j) == "Cannot solve opcode: expression has too many unknowns " + expr_display([], lin, qc), j
        # ... and as data (acvm_batch_error_expression): what a binding rebuilds OpcodeNotSolvable::ExpressionHasTooManyUnknowns(Expression) from
        assert b.error_expression(j) == {"opcode_index": 0, "mul": [], "lin": lin, "q_c": qc}, j
    # two unknown multiplicands stay a mul term; a bare unknown witness prints as x{w}
    circ = Circuit(9, [E([(P - 2, 8, 9)], [(1, 1)], 0)])
    c, b = solve(circ, [1], [[5]])
    assert b.error_string(0) == "Cannot solve opcode: expression has too many unknowns " + expr_display([(P - 2, 8, 9)], [], 5)
    assert b.error_expression(0) == {"opcode_index": 0, "mul": [(P - 2, 8, 9)], "lin": [], "q_c": 5}
    # Brillig: the input expression as written (brillig.rs:46-74), the first one that does not reduce to a constant
    br = Brillig(inputs=[E.from_witness(1), [E([], [(1, 1), (3, 9)], 4), E.from_witness(8)], E.from_witness(8)], outputs=[7], bytecode=[("Stop",)])
    circ = Circuit(9, [br])
    c, b = solve(circ, [1], [[5]])
    assert b.error_string(0) == "Cannot solve opcode: expression has too many unknowns " + expr_display([], [(1, 1), (3, 9)], 4)
    assert b.error_expression(0) == {"opcode_index": 0, "mul": [], "lin": [(1, 1), (3, 9)], "q_c": 4}
    br = Brillig(inputs=[E.from_witness(8)], outputs=[7], bytecode=[("Stop",)])
    c, b = solve(Circuit(9, [br]), [1], [[5]])
    assert b.error_string(0) == "Cannot solve opcode: expression has too many unknowns x8"
    assert b.error_expression(0) == {"opcode_index": 0, "mul": [], "lin": [(1, 8)], "q_c": 0}
    # an instance that solved (or failed otherwise) carries no such expression; ACVM::opcodes through the ABI
    c, b = solve(Circuit(3, [E([], [(1, 1), (P - 1, 2)], 0), E([], [(1, 2)], 0)]), [1], [[0], [5]])
    assert b.error_expression(0) is None and b.error_expression(1) is None and b.results()[1].err == acvm_amd_err("UNSATISFIED")


def acvm_amd_err(name):
    import acvm_amd
    return getattr(acvm_amd, "ERR_" + name)


def test_opcode_kinds_mirror_the_circuit():
    """acvm_circuit_opcode_kinds: ACVM::opcodes (pwg/mod.rs:166-168) as (variant, sub-kind) per opcode, for the seven reference circuits' opcode mix"""
    import acvm_amd
    from acvm_amd import synth
    circ, ids = synth.mixed_circuit(400, seed=0x0C0DE)
    gc = acvm_amd.Circuit(circ.to_bytes())
    kinds = gc.opcode_kinds()
    assert len(kinds) == len(circ.opcodes) == gc.num_opcodes
    names = {"Expression": (0, 0), "BlackBoxFuncCall": (1, None), "QuotientDirective": (2, 0), "ToLeRadix": (2, 1), "PermutationSort": (2, 2),
             "Brillig": (3, None), "MemoryOp": (4, None), "MemoryInit": (5, None)}
    assert len({k for k, _ in kinds}) >= 5  # the mix holds most variants
    for (k, sub), op in zip(kinds, circ.opcodes):
        want_k, want_sub = names[type(op).__name__]
        assert k == want_k and (want_sub is None or sub == want_sub), (k, sub, type(op).__name__)
        if k == 3:
            assert sub == len(op.bytecode)
        if k in (4, 5):
            assert sub == op.block_id
