// circuit.hpp -- host-side ACIR circuit model of the product and its wire-format reader.
// Mirrors acir::circuit::{Circuit, Opcode, ...} (acir/src/circuit/mod.rs:18-41, opcodes.rs:15-34,
// opcodes/black_box_function_call.rs:20-115, directives.rs:4-46, brillig.rs:8-33,
// native_types/expression/mod.rs:17-28, brillig/src/opcodes.rs:60-134). Reader = Circuit::read
// (circuit/mod.rs:154-161): gzip + bincode 1.3 default config.
#pragma once
#include "fr_host.hpp"
#include <memory>
#include <string>
#include <vector>

namespace acvm {

struct MulTerm { FrH c; uint32_t l, r; };
struct LinTerm { FrH c; uint32_t w; };
struct Expr {
    std::vector<MulTerm> mul;
    std::vector<LinTerm> lin;
    FrH qc = frh::zero();
};
struct FuncInput { uint32_t witness, num_bits; };

enum BlackBoxTag : uint32_t {
    BB_AND = 0, BB_XOR, BB_RANGE, BB_SHA256, BB_BLAKE2S, BB_SCHNORR_VERIFY, BB_PEDERSEN, BB_HASH_TO_FIELD_128,
    BB_ECDSA_SECP256K1, BB_ECDSA_SECP256R1, BB_FIXED_BASE_SCALAR_MUL, BB_KECCAK256, BB_KECCAK256_VAR,
    BB_RECURSIVE_AGGREGATION, BB_COUNT
};
struct BlackBoxCall {
    uint32_t func = 0;
    std::vector<FuncInput> in[4];  // input groups in declaration order (see get_inputs_vec, :205-292)
    uint32_t domain_separator = 0;
    std::vector<uint32_t> out;
    bool has_in_agg = false;
    std::vector<FuncInput> in_agg;
};

enum DirectiveKind : uint32_t { DIR_QUOTIENT = 0, DIR_TO_LE_RADIX = 1, DIR_PERMUTATION_SORT = 2 };
struct Directive {
    uint32_t kind = 0;
    Expr a, b;
    uint32_t q = 0, r = 0;
    bool has_predicate = false;
    Expr predicate;
    std::vector<uint32_t> bw;
    uint32_t radix = 0;
    std::vector<std::vector<Expr>> sort_inputs;
    uint32_t tuple = 0;
    std::vector<uint32_t> sort_by;
};

enum BrilligOpKind : uint32_t {
    BR_BINARY_FIELD_OP = 0, BR_BINARY_INT_OP, BR_JUMP_IF_NOT, BR_JUMP_IF, BR_JUMP, BR_CALL, BR_CONST, BR_RETURN,
    BR_FOREIGN_CALL, BR_MOV, BR_LOAD, BR_STORE, BR_BLACK_BOX, BR_TRAP, BR_STOP
};
struct RegOrMem { uint32_t kind; uint64_t reg, size; };
struct BrilligOp {
    uint32_t op = 0;
    uint64_t a = 0, b = 0, c = 0;
    uint32_t sub_op = 0, bit_size = 0;
    uint64_t location = 0;
    FrH value = frh::zero();
    std::string function;
    std::vector<RegOrMem> dests, inputs;
    uint32_t bbop = 0;
    uint64_t bb[10] = {0};
};
struct FcOutput { bool is_array = false; FrH single = frh::zero(); std::vector<FrH> arr; };
struct FcResult { std::vector<FcOutput> values; };
struct BrilligInput { bool is_array = false; Expr single; std::vector<Expr> arr; };
struct BrilligOutput { bool is_array = false; uint32_t w = 0; std::vector<uint32_t> arr; };
struct BrilligCall {
    std::vector<BrilligInput> inputs;
    std::vector<BrilligOutput> outputs;
    std::vector<FcResult> fc_results;
    std::vector<BrilligOp> bytecode;
    bool has_predicate = false;
    Expr predicate;
};

enum OpcodeKind : uint32_t { OP_ARITHMETIC = 0, OP_BLACKBOX = 1, OP_DIRECTIVE = 2, OP_BRILLIG = 3, OP_MEMORY_OP = 4, OP_MEMORY_INIT = 5 };
struct Opcode {
    uint32_t kind = 0;
    Expr expr;  // Arithmetic
    std::unique_ptr<BlackBoxCall> bb;
    std::unique_ptr<Directive> dir;
    std::unique_ptr<BrilligCall> brillig;
    uint32_t block_id = 0;
    Expr mem_operation, mem_index, mem_value;  // MemoryOp
    bool has_predicate = false;
    Expr predicate;
    std::vector<uint32_t> init;  // MemoryInit
};

struct AssertMessage { bool is_brillig; uint64_t acir_index, brillig_index; std::string message; };

struct Circuit {
    uint32_t current_witness_index = 0;
    std::vector<Opcode> opcodes;
    std::vector<uint32_t> private_parameters, public_parameters, return_values;
    std::vector<AssertMessage> assert_messages;
    uint32_t max_witness = 0;
};

// Circuit::read. Accepts gzip(bincode) or raw bincode. Returns nullptr and fills err on malformed input.
std::unique_ptr<Circuit> circuit_from_bytes(const uint8_t *buf, size_t len, std::string &err);

// WitnessMap <-> bytes (acir/src/native_types/witness_map.rs:108-146): gzip(bincode(BTreeMap<Witness(u32), FieldElement>)),
// a FieldElement being its 64-character hex string. Values are canonical 32-byte big-endian (reduced on read).
bool witness_map_from_bytes(const uint8_t *buf, size_t len, std::vector<uint32_t> &ids, std::vector<uint8_t> &values_be32, std::string &err);
bool witness_map_to_bytes(const uint32_t *ids, const uint8_t *values_be32, size_t n, std::vector<uint8_t> &out, std::string &err);

}  // namespace acvm
