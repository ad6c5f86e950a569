"""Wire format parity: the Python writer (acvm_amd/acir.py), the oracle's reader and the product's reader against the
seven byte-exact circuits of acir/tests/test_program_serialization.rs (tests/golden/reference_vectors.json)."""
import gzip

import pytest

from acvm_amd import acir
from acvm_amd.acir import (Arithmetic, BlackBoxFuncCall, Brillig, Circuit, Expression, FunctionInput, MemoryInit,
                           MemoryOp, P)


def _ref_circuits():
    one, m1 = 1, P - 1
    out = {}
    # acir/tests/test_program_serialization.rs:25-60
    out["addition_circuit"] = Circuit(4, [Arithmetic(Expression([], [(one, 1), (one, 2), (m1, 3)], 0))],
                                      private_parameters=[1, 2], return_values=[3])
    # :62-85
    out["fixed_base_scalar_mul_circuit"] = Circuit(
        5, [BlackBoxFuncCall("FixedBaseScalarMul", dict(low=FunctionInput(1, 128), high=FunctionInput(2, 128), outputs=(3, 4)))],
        private_parameters=[1, 2], return_values=[3, 4])
    # :87-112
    out["pedersen_circuit"] = Circuit(
        4, [BlackBoxFuncCall("Pedersen", dict(inputs=[FunctionInput(1, 254)], outputs=(2, 3), domain_separator=0))],
        private_parameters=[1], return_values=[2, 3])
    # :114-161
    sig = [FunctionInput(i, 8) for i in range(3, 3 + 64)]
    msg = [FunctionInput(i, 8) for i in range(3 + 64, 3 + 74)]
    out["schnorr_verify_circuit"] = Circuit(
        100, [BlackBoxFuncCall("SchnorrVerify", dict(public_key_x=FunctionInput(1, 254), public_key_y=FunctionInput(2, 254),
                                                     signature=sig, message=msg, output=77))],
        private_parameters=list(range(1, 77)), return_values=[77])
    # :163-208
    w_in, w_inv = 1, 2
    out["simple_brillig_foreign_call"] = Circuit(
        8, [Brillig(inputs=[Expression([], [(one, w_in)], 0)], outputs=[w_inv],
                    bytecode=[("ForeignCall", "invert", [("Register", 0)], [("Register", 0)])])],
        private_parameters=[w_in, w_inv])
    # :210-285
    a, b, c, ax2, bx2, cx2, sm, prod = 1, 2, 3, 4, 5, 6, 7, 8
    out["complex_brillig_foreign_call"] = Circuit(
        8, [Brillig(
            inputs=[[Expression.from_witness(a), Expression.from_witness(b), Expression.from_witness(c)],
                    Expression([], [(one, a), (one, b), (one, c)], 0)],
            outputs=[[ax2, bx2, cx2], sm, prod],
            bytecode=[("ForeignCall", "complex",
                       [("HeapArray", 0, 3), ("Register", 1), ("Register", 2)],
                       [("HeapArray", 0, 3), ("Register", 1)])])],
        private_parameters=[a, b, c])
    # :287-330
    out["memory_op_circuit"] = Circuit(
        5, [MemoryInit(0, [1, 2]),
            MemoryOp(0, Expression.constant(1), Expression.constant(1), Expression.from_witness(3)),
            MemoryOp(0, Expression.constant(0), Expression.constant(1), Expression.from_witness(4))],
        private_parameters=[1, 2, 3], return_values=[4])
    return out


@pytest.mark.parametrize("name", ["addition_circuit", "fixed_base_scalar_mul_circuit", "pedersen_circuit",
                                  "schnorr_verify_circuit", "simple_brillig_foreign_call",
                                  "complex_brillig_foreign_call", "memory_op_circuit"])
def test_python_writer_is_byte_exact(golden, name):
    ref = bytes(golden["serialization"][name])
    mine = _ref_circuits()[name]
    assert mine.to_bincode() == gzip.decompress(ref)


def test_oracle_reader_accepts_all_reference_circuits(oracle, golden):
    expect = {"addition_circuit": (1, 5), "fixed_base_scalar_mul_circuit": (1, 6), "pedersen_circuit": (1, 5),
              "schnorr_verify_circuit": (1, 101), "simple_brillig_foreign_call": (1, 9),
              "complex_brillig_foreign_call": (1, 9), "memory_op_circuit": (3, 6)}
    for name, data in golden["serialization"].items():
        c = oracle.Circuit(bytes(data))
        assert (c.num_opcodes, c.num_witnesses) == expect[name], name
        c2 = oracle.Circuit(gzip.decompress(bytes(data)))  # raw bincode is accepted too
        assert (c2.num_opcodes, c2.num_witnesses) == expect[name]


def test_product_reader_accepts_all_reference_circuits(golden):
    import acvm_amd
    for name, data in golden["serialization"].items():
        c = acvm_amd.Circuit(bytes(data))
        assert c.num_opcodes >= 1


def test_readers_reject_malformed(oracle, golden):
    import acvm_amd
    raw = gzip.decompress(bytes(golden["serialization"]["addition_circuit"]))
    for bad in [raw[:-1], raw + b"\x00", raw[:10], b"\x1f\x8b" + raw, b""]:
        with pytest.raises(ValueError):
            oracle.Circuit(bad)
        with pytest.raises(acvm_amd.AcvmError):
            acvm_amd.Circuit(bad)


def test_field_hex_is_reduced_like_from_hex():
    # generic_ark.rs:263-267: from_hex reduces mod p. The writer always emits canonical residues.
    e = Expression([], [(P + 5, 1)], P)
    assert acir._fr(P + 5) == acir._fr(5)
    assert e.q_c == P
