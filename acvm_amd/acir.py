"""ACIR circuit data model (host side, Python) and its bincode wire format.

Mirrors the reference's IR types so tests and the bench can express the same circuits the Rust API takes:
  acir/src/circuit/mod.rs:18-41            Circuit
  acir/src/circuit/opcodes.rs:15-34        Opcode
  acir/src/circuit/opcodes/black_box_function_call.rs:20-115  BlackBoxFuncCall (tags 0..13)
  acir/src/circuit/directives.rs:4-46      Directive
  acir/src/circuit/brillig.rs:8-33         Brillig, BrilligInputs/Outputs
  acir/src/native_types/expression/mod.rs:17-28  Expression
  brillig/src/opcodes.rs:60-134            Brillig bytecode
Wire format = gzip(bincode 1.3 default config) as written by Circuit::write (circuit/mod.rs:145-151):
fixint little-endian, u64 lengths, u32 enum tags, u8 Option tag, FieldElement = 64-char hex String.
Field elements are plain Python ints (canonical residues mod P).
"""
from __future__ import annotations

import gzip
import struct
from dataclasses import dataclass, field
from typing import List, Optional, Tuple, Union

P = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001  # BN254 Fr


def fe(x: int) -> int:
    return x % P


# --------------------------------------------------------------------------- model
@dataclass
class Expression:
    mul_terms: List[Tuple[int, int, int]] = field(default_factory=list)  # (q_m, w_l, w_r)
    linear_combinations: List[Tuple[int, int]] = field(default_factory=list)  # (q, w)
    q_c: int = 0

    @staticmethod
    def from_witness(w: int) -> "Expression":
        return Expression([], [(1, w)], 0)

    @staticmethod
    def constant(c: int) -> "Expression":
        return Expression([], [], fe(c))


@dataclass
class FunctionInput:
    witness: int
    num_bits: int


@dataclass
class Arithmetic:
    expr: Expression


@dataclass
class BlackBoxFuncCall:
    """name in {AND, XOR, RANGE, SHA256, Blake2s, SchnorrVerify, Pedersen, HashToField128Security,
    EcdsaSecp256k1, EcdsaSecp256r1, FixedBaseScalarMul, Keccak256, Keccak256VariableLength,
    RecursiveAggregation}; args follow the Rust field order."""
    name: str
    args: dict


@dataclass
class QuotientDirective:
    a: Expression
    b: Expression
    q: int
    r: int
    predicate: Optional[Expression] = None


@dataclass
class ToLeRadix:
    a: Expression
    b: List[int]
    radix: int


@dataclass
class PermutationSort:
    inputs: List[List[Expression]]
    tuple: int
    bits: List[int]
    sort_by: List[int]


@dataclass
class MemoryOp:
    block_id: int
    operation: Expression
    index: Expression
    value: Expression
    predicate: Optional[Expression] = None


@dataclass
class MemoryInit:
    block_id: int
    init: List[int]


# brillig bytecode: tuples ("BinaryFieldOp", dest, op, lhs, rhs) etc. — see _ser_brillig_op
BINARY_FIELD_OPS = ["Add", "Sub", "Mul", "Div", "Equals"]
BINARY_INT_OPS = ["Add", "Sub", "Mul", "SignedDiv", "UnsignedDiv", "Equals", "LessThan", "LessThanEquals",
                  "And", "Or", "Xor", "Shl", "Shr"]


@dataclass
class Brillig:
    inputs: List[Union[Expression, List[Expression]]]  # Single(expr) | Array([expr])
    outputs: List[Union[int, List[int]]]  # Simple(w) | Array([w])
    bytecode: List[tuple]
    foreign_call_results: List[List[Union[int, List[int]]]] = field(default_factory=list)
    predicate: Optional[Expression] = None


@dataclass
class Circuit:
    current_witness_index: int
    opcodes: list
    private_parameters: List[int] = field(default_factory=list)
    public_parameters: List[int] = field(default_factory=list)
    return_values: List[int] = field(default_factory=list)
    assert_messages: List[Tuple[Union[int, Tuple[int, int]], str]] = field(default_factory=list)

    def to_bincode(self) -> bytes:
        return _ser_circuit(self)

    def to_bytes(self) -> bytes:
        """Circuit::write: gzip(bincode)."""
        return gzip.compress(self.to_bincode(), compresslevel=6, mtime=0)


BLACKBOX_TAGS = ["AND", "XOR", "RANGE", "SHA256", "Blake2s", "SchnorrVerify", "Pedersen", "HashToField128Security",
                 "EcdsaSecp256k1", "EcdsaSecp256r1", "FixedBaseScalarMul", "Keccak256", "Keccak256VariableLength",
                 "RecursiveAggregation"]


# --------------------------------------------------------------------------- bincode writer
def _u8(v):
    return struct.pack("<B", v)


def _u32(v):
    return struct.pack("<I", v)


def _u64(v):
    return struct.pack("<Q", v)


def _fr(x: int) -> bytes:
    s = ("%064x" % (x % P)).encode()
    return _u64(64) + s


def _str(s: str) -> bytes:
    b = s.encode()
    return _u64(len(b)) + b


def _ser_expr(e: Expression) -> bytes:
    out = [_u64(len(e.mul_terms))]
    for c, l, r in e.mul_terms:
        out += [_fr(c), _u32(l), _u32(r)]
    out.append(_u64(len(e.linear_combinations)))
    for c, w in e.linear_combinations:
        out += [_fr(c), _u32(w)]
    out.append(_fr(e.q_c))
    return b"".join(out)


def _opt_expr(e: Optional[Expression]) -> bytes:
    return _u8(0) if e is None else _u8(1) + _ser_expr(e)


def _fi(f: FunctionInput) -> bytes:
    return _u32(f.witness) + _u32(f.num_bits)


def _fi_vec(v) -> bytes:
    return _u64(len(v)) + b"".join(_fi(f) for f in v)


def _w_vec(v) -> bytes:
    return _u64(len(v)) + b"".join(_u32(w) for w in v)


def _ser_blackbox(b: BlackBoxFuncCall) -> bytes:
    tag = BLACKBOX_TAGS.index(b.name)
    a = b.args
    out = [_u32(tag)]
    n = b.name
    if n in ("AND", "XOR"):
        out += [_fi(a["lhs"]), _fi(a["rhs"]), _u32(a["output"])]
    elif n == "RANGE":
        out += [_fi(a["input"])]
    elif n in ("SHA256", "Blake2s", "Keccak256"):
        out += [_fi_vec(a["inputs"]), _w_vec(a["outputs"])]
    elif n == "SchnorrVerify":
        out += [_fi(a["public_key_x"]), _fi(a["public_key_y"]), _fi_vec(a["signature"]), _fi_vec(a["message"]),
                _u32(a["output"])]
    elif n == "Pedersen":
        out += [_fi_vec(a["inputs"]), _u32(a["domain_separator"]), _u32(a["outputs"][0]), _u32(a["outputs"][1])]
    elif n == "HashToField128Security":
        out += [_fi_vec(a["inputs"]), _u32(a["output"])]
    elif n in ("EcdsaSecp256k1", "EcdsaSecp256r1"):
        out += [_fi_vec(a["public_key_x"]), _fi_vec(a["public_key_y"]), _fi_vec(a["signature"]),
                _fi_vec(a["hashed_message"]), _u32(a["output"])]
    elif n == "FixedBaseScalarMul":
        out += [_fi(a["low"]), _fi(a["high"]), _u32(a["outputs"][0]), _u32(a["outputs"][1])]
    elif n == "Keccak256VariableLength":
        out += [_fi_vec(a["inputs"]), _fi(a["var_message_size"]), _w_vec(a["outputs"])]
    elif n == "RecursiveAggregation":
        out += [_fi_vec(a["verification_key"]), _fi_vec(a["proof"]), _fi_vec(a["public_inputs"]), _fi(a["key_hash"])]
        agg = a.get("input_aggregation_object")
        out += [_u8(0)] if agg is None else [_u8(1), _fi_vec(agg)]
        out += [_w_vec(a["output_aggregation_object"])]
    else:  # pragma: no cover
        raise ValueError(n)
    return b"".join(out)


def _rom(x) -> bytes:
    """RegisterOrMemory: ("Register", r) | ("HeapArray", ptr, size) | ("HeapVector", ptr, size_reg)."""
    if x[0] == "Register":
        return _u32(0) + _u64(x[1])
    if x[0] == "HeapArray":
        return _u32(1) + _u64(x[1]) + _u64(x[2])
    return _u32(2) + _u64(x[1]) + _u64(x[2])


_BB_OPS = ["Sha256", "Blake2s", "Keccak256", "HashToField128Security", "EcdsaSecp256k1", "EcdsaSecp256r1",
           "SchnorrVerify", "Pedersen", "FixedBaseScalarMul"]


def _ser_brillig_op(o: tuple) -> bytes:
    k = o[0]
    if k == "BinaryFieldOp":  # (k, destination, op, lhs, rhs)
        return _u32(0) + _u64(o[1]) + _u32(BINARY_FIELD_OPS.index(o[2])) + _u64(o[3]) + _u64(o[4])
    if k == "BinaryIntOp":  # (k, destination, op, bit_size, lhs, rhs)
        return _u32(1) + _u64(o[1]) + _u32(BINARY_INT_OPS.index(o[2])) + _u32(o[3]) + _u64(o[4]) + _u64(o[5])
    if k == "JumpIfNot":
        return _u32(2) + _u64(o[1]) + _u64(o[2])
    if k == "JumpIf":
        return _u32(3) + _u64(o[1]) + _u64(o[2])
    if k == "Jump":
        return _u32(4) + _u64(o[1])
    if k == "Call":
        return _u32(5) + _u64(o[1])
    if k == "Const":  # (k, destination, value)
        return _u32(6) + _u64(o[1]) + _fr(o[2])
    if k == "Return":
        return _u32(7)
    if k == "ForeignCall":  # (k, function, destinations, inputs)
        return (_u32(8) + _str(o[1]) + _u64(len(o[2])) + b"".join(_rom(x) for x in o[2]) + _u64(len(o[3]))
                + b"".join(_rom(x) for x in o[3]))
    if k == "Mov":  # (k, destination, source)
        return _u32(9) + _u64(o[1]) + _u64(o[2])
    if k == "Load":  # (k, destination, source_pointer)
        return _u32(10) + _u64(o[1]) + _u64(o[2])
    if k == "Store":  # (k, destination_pointer, source)
        return _u32(11) + _u64(o[1]) + _u64(o[2])
    if k == "BlackBox":  # (k, name, *u64 words in declaration order)
        return _u32(12) + _u32(_BB_OPS.index(o[1])) + b"".join(_u64(w) for w in o[2:])
    if k == "Trap":
        return _u32(13)
    if k == "Stop":
        return _u32(14)
    raise ValueError(k)


def _ser_brillig(b: Brillig) -> bytes:
    out = [_u64(len(b.inputs))]
    for i in b.inputs:
        if isinstance(i, Expression):
            out += [_u32(0), _ser_expr(i)]
        else:
            out += [_u32(1), _u64(len(i))] + [_ser_expr(e) for e in i]
    out.append(_u64(len(b.outputs)))
    for o in b.outputs:
        if isinstance(o, int):
            out += [_u32(0), _u32(o)]
        else:
            out += [_u32(1), _w_vec(o)]
    out.append(_u64(len(b.foreign_call_results)))
    for r in b.foreign_call_results:
        out.append(_u64(len(r)))
        for v in r:
            if isinstance(v, int):
                out += [_u32(0), _fr(v)]
            else:
                out += [_u32(1), _u64(len(v))] + [_fr(x) for x in v]
    out.append(_u64(len(b.bytecode)))
    out += [_ser_brillig_op(o) for o in b.bytecode]
    out.append(_opt_expr(b.predicate))
    return b"".join(out)


def _ser_opcode(o) -> bytes:
    if isinstance(o, Arithmetic):
        return _u32(0) + _ser_expr(o.expr)
    if isinstance(o, Expression):
        return _u32(0) + _ser_expr(o)
    if isinstance(o, BlackBoxFuncCall):
        return _u32(1) + _ser_blackbox(o)
    if isinstance(o, QuotientDirective):
        return (_u32(2) + _u32(0) + _ser_expr(o.a) + _ser_expr(o.b) + _u32(o.q) + _u32(o.r) + _opt_expr(o.predicate))
    if isinstance(o, ToLeRadix):
        return _u32(2) + _u32(1) + _ser_expr(o.a) + _w_vec(o.b) + _u32(o.radix)
    if isinstance(o, PermutationSort):
        out = [_u32(2), _u32(2), _u64(len(o.inputs))]
        for t in o.inputs:
            out += [_u64(len(t))] + [_ser_expr(e) for e in t]
        out += [_u32(o.tuple), _w_vec(o.bits), _u64(len(o.sort_by))] + [_u32(x) for x in o.sort_by]
        return b"".join(out)
    if isinstance(o, Brillig):
        return _u32(3) + _ser_brillig(o)
    if isinstance(o, MemoryOp):
        return (_u32(4) + _u32(o.block_id) + _ser_expr(o.operation) + _ser_expr(o.index) + _ser_expr(o.value)
                + _opt_expr(o.predicate))
    if isinstance(o, MemoryInit):
        return _u32(5) + _u32(o.block_id) + _w_vec(o.init)
    raise TypeError(type(o))


def _ser_circuit(c: Circuit) -> bytes:
    out = [_u32(c.current_witness_index), _u64(len(c.opcodes))]
    out += [_ser_opcode(o) for o in c.opcodes]
    out += [_w_vec(sorted(set(c.private_parameters))), _w_vec(sorted(set(c.public_parameters))),
            _w_vec(sorted(set(c.return_values)))]
    out.append(_u64(len(c.assert_messages)))
    for loc, msg in c.assert_messages:
        if isinstance(loc, int):
            out += [_u32(0), _u64(loc)]
        else:
            out += [_u32(1), _u64(loc[0]), _u64(loc[1])]
        out.append(_str(msg))
    return b"".join(out)
