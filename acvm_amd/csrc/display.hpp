// display.hpp -- the reference's Display texts of field elements and expressions, for error strings that quote them
// (OpcodeNotSolvable::ExpressionHasTooManyUnknowns(Expression), acvm/src/pwg/mod.rs:72-78). Host only.
#pragma once
#include "circuit.hpp"
#include <string>

namespace acvm {

// impl Display for FieldElement (acir_field/src/generic_ark.rs:13-74): the shorter of x and -(p - x) in decimal, powers of two as
// 2 with a superscript exponent, multiples of 2^64 / 2^32 / 2^16 / 2^8 / 2^4 as "2^k x q"
std::string field_display(const FrH &x);
// impl Display for Expression (acir/src/native_types/expression/mod.rs:40-48): "x{w}" for a bare witness, else the Debug text of
// Opcode::Arithmetic between percent signs: "%EXPR [ (c, _l, _r) ... (c, _w) ... q_c ]%" (acir/src/circuit/opcodes.rs:88-102)
std::string expression_display(const Expr &e);

}  // namespace acvm
