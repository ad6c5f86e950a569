"""Circuit bytes are untrusted input: nothing may wrap, over-allocate or unwind through the C ABI.
Host-only: acvm_circuit_from_bytes / acvm_circuit_plan_stats need no device."""
import gzip

import pytest

import acvm_amd
from acvm_amd.acir import P, Circuit, Expression as E, PermutationSort


def test_witness_index_near_2_32_is_refused_not_wrapped():
    for top in (0xFFFFFFFF, 0x7FFFFFF0, 1 << 27):
        circ = Circuit(4, [E([], [(1, 1), (P - 1, top)], 0)])
        c = acvm_amd.Circuit(circ.to_bytes())
        with pytest.raises(acvm_amd.AcvmError, match="dense witness table"):
            c.plan_stats([1])
    # the largest index the dense table takes is still planned
    circ = Circuit(4, [E([], [(1, 1), (P - 1, 1000000)], 0)])
    assert acvm_amd.Circuit(circ.to_bytes()).plan_stats([1])["n_witnesses"] == 1000001
    # ... and an initial witness id beyond it is refused the same way
    with pytest.raises(acvm_amd.AcvmError, match="dense witness table"):
        acvm_amd.Circuit(Circuit(4, [E([], [(1, 1), (P - 1, 2)], 0)]).to_bytes()).plan_stats([1, 0xFFFFFFFF])


def test_field_element_hex_of_any_even_length_is_reduced(oracle):
    """from_hex = hex::decode + from_be_bytes_reduce (generic_ark.rs:263-283): 66 hex digits are a valid coefficient"""
    import struct
    circ = Circuit(2, [E([], [(5, 1), (P - 1, 2)], 0)])
    raw = gzip.decompress(circ.to_bytes())
    short = b"%064x" % 5
    pos = raw.index(short)
    big = (7 << 256) + 5  # 33 bytes
    long_hex = b"%066x" % big
    patched = raw[:pos - 8] + struct.pack("<Q", 66) + long_hex + raw[pos + 64:]
    c = acvm_amd.Circuit(patched)
    assert c.num_opcodes == 1 and c.plan_stats([1])["n_fast_gates"] == 1
    # the oracle reads the same bytes and agrees on the value: w2 = (big mod p) * w1
    a = oracle.ACVM(oracle.Circuit(patched), {1: 3})
    assert a.solve() == oracle.ST_SOLVED and a.witness_map()[2] == (big % P) * 3 % P
    assert int.from_bytes(oracle_reduce(oracle, big.to_bytes(33, "big")), "big") == big % P


def oracle_reduce(oracle, data):
    import ctypes as C
    out = C.create_string_buffer(32)
    oracle.lib().oracle_fr_from_bytes_reduce(data, len(data), out)
    return out.raw


def test_same_unknown_in_mul_and_linear_term_is_too_many_unknowns(oracle):
    """q*x*w + b*w with x known: evaluate() turns the product into a second linear term on w, solve_fan_in_term counts two
    unknowns (arithmetic.rs:188-201,212-239) -> ExpressionHasTooManyUnknowns; the `w1 == w2` arm of solve is unreachable behind
    evaluate. The planner therefore hands the opcode to the exact kernels (truncated_at), it is not a solvable gate."""
    circ = Circuit(3, [E([(2, 1, 2)], [(3, 2)], 5)])
    a = oracle.ACVM(oracle.Circuit(circ.to_bytes()), {1: 7})
    assert a.solve() == oracle.ST_FAILURE and a.result().err == oracle.E_TOO_MANY_UNKNOWNS
    assert acvm_amd.Circuit(circ.to_bytes()).plan_stats([1])["truncated_at"] == 0
    # x == 0 drops the product (zero coefficient, arithmetic.rs:217-221): then the opcode IS solvable
    b = oracle.ACVM(oracle.Circuit(circ.to_bytes()), {1: 0})
    assert b.solve() == oracle.ST_SOLVED and b.witness_map()[2] == (-5 * pow(3, -1, P)) % P


def test_sort_column_beyond_the_tuple_panics_in_the_oracle(oracle):
    """a[*i as usize] on a Vec of tuple + 1 elements (directives/mod.rs:102-105): index out of bounds once two elements are compared"""
    w = E.from_witness
    circ = Circuit(8, [PermutationSort([[w(1)], [w(2)], [w(3)]], 1, [4, 5, 6], [2])])
    a = oracle.ACVM(oracle.Circuit(circ.to_bytes()), {1: 3, 2: 2, 3: 1})
    assert a.solve() == oracle.ST_FAILURE
    r = a.result()
    assert r.err == oracle.E_PANIC and r.message == b"index out of bounds: the len is 2 but the index is 2"
    # column == tuple is the element's own index: fine
    ok = Circuit(8, [PermutationSort([[w(1)], [w(2)], [w(3)]], 1, [4, 5, 6], [1])])
    assert oracle.ACVM(oracle.Circuit(ok.to_bytes()), {1: 3, 2: 2, 3: 1}).solve() == oracle.ST_SOLVED
