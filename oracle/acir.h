/*
 * oracle/acir.h -- TEST INFRASTRUCTURE ONLY (CPU oracle). Not part of the shipped product path.
 *
 * Plain-C data model of the ACIR circuit, restating
 *   /root/reference/acir/src/circuit/mod.rs:18-41           (Circuit)
 *   /root/reference/acir/src/circuit/opcodes.rs:15-34       (Opcode)
 *   /root/reference/acir/src/circuit/opcodes/black_box_function_call.rs:20-115 (BlackBoxFuncCall)
 *   /root/reference/acir/src/circuit/directives.rs:4-46     (Directive)
 *   /root/reference/acir/src/circuit/brillig.rs:8-33        (Brillig, inputs/outputs)
 *   /root/reference/acir/src/circuit/opcodes/memory_operation.rs:4-16 (MemOp)
 *   /root/reference/acir/src/native_types/expression/mod.rs:17-28 (Expression)
 *   /root/reference/brillig/src/{opcodes,black_box,foreign_call,value}.rs (Brillig bytecode)
 * and a reader for the bincode-1.3 default encoding used by Circuit::write (circuit/mod.rs:145-161;
 * layout observed in SURVEY Appendix D). Pinned by the seven byte-exact circuits of
 * acir/tests/test_program_serialization.rs (tests/golden/serialization_*.json).
 */
#ifndef ORACLE_ACIR_H
#define ORACLE_ACIR_H
#include "fr.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct { fr_t c; uint32_t l, r; } mul_term_t;
typedef struct { fr_t c; uint32_t w; } lin_term_t;
typedef struct {
    mul_term_t *mul; size_t n_mul;
    lin_term_t *lin; size_t n_lin;
    fr_t qc;
} expr_t;

typedef struct { uint32_t witness, num_bits; } func_input_t;

/* BlackBoxFuncCall tags in declaration order (black_box_function_call.rs:20-115) */
enum {
    BB_AND = 0, BB_XOR, BB_RANGE, BB_SHA256, BB_BLAKE2S, BB_SCHNORR_VERIFY, BB_PEDERSEN,
    BB_HASH_TO_FIELD_128, BB_ECDSA_SECP256K1, BB_ECDSA_SECP256R1, BB_FIXED_BASE_SCALAR_MUL,
    BB_KECCAK256, BB_KECCAK256_VAR, BB_RECURSIVE_AGGREGATION, BB_COUNT
};

typedef struct {
    uint32_t func;
    /* input groups in declaration order:
     *  AND/XOR: in[0]={lhs}, in[1]={rhs}; RANGE: in[0]={input}; hashes/pedersen/h2f: in[0]=inputs;
     *  schnorr: in[0]={pkx}, in[1]={pky}, in[2]=signature, in[3]=message;
     *  ecdsa: in[0]=pkx, in[1]=pky, in[2]=signature, in[3]=hashed_message;
     *  fixed base: in[0]={low}, in[1]={high}; keccak var: in[0]=inputs, in[1]={var_message_size};
     *  recursion: in[0]=vk, in[1]=proof, in[2]=public_inputs, in[3]={key_hash} */
    func_input_t *in[4]; size_t n_in[4];
    uint32_t domain_separator;
    uint32_t *out; size_t n_out;
    int has_in_agg; func_input_t *in_agg; size_t n_in_agg;
} bb_call_t;

enum { DIR_QUOTIENT = 0, DIR_TO_LE_RADIX = 1, DIR_PERMUTATION_SORT = 2 };
typedef struct {
    uint32_t kind;
    expr_t a, b;            /* quotient: a,b ; to_le_radix: a */
    uint32_t q, r;          /* quotient */
    int has_predicate; expr_t predicate;
    uint32_t *bw; size_t n_bw; /* to_le_radix outputs b ; permutation sort bits */
    uint32_t radix;
    /* permutation sort */
    expr_t **sort_inputs; size_t *sort_input_len; size_t n_sort_inputs;
    uint32_t tuple; uint32_t *sort_by; size_t n_sort_by;
} directive_t;

/* ---- brillig bytecode (brillig/src/opcodes.rs:60-134) ---- */
enum {
    BR_BINARY_FIELD_OP = 0, BR_BINARY_INT_OP, BR_JUMP_IF_NOT, BR_JUMP_IF, BR_JUMP, BR_CALL, BR_CONST,
    BR_RETURN, BR_FOREIGN_CALL, BR_MOV, BR_LOAD, BR_STORE, BR_BLACK_BOX, BR_TRAP, BR_STOP
};
enum { BF_ADD = 0, BF_SUB, BF_MUL, BF_DIV, BF_EQUALS };
enum { BI_ADD = 0, BI_SUB, BI_MUL, BI_SIGNED_DIV, BI_UNSIGNED_DIV, BI_EQUALS, BI_LT, BI_LTE, BI_AND, BI_OR, BI_XOR, BI_SHL, BI_SHR };
enum { ROM_REGISTER = 0, ROM_HEAP_ARRAY = 1, ROM_HEAP_VECTOR = 2 };
typedef struct { uint32_t kind; uint64_t reg; uint64_t size; /* array: literal size; vector: size register */ } reg_or_mem_t;
enum { BBOP_SHA256 = 0, BBOP_BLAKE2S, BBOP_KECCAK256, BBOP_HASH_TO_FIELD, BBOP_ECDSA_K1, BBOP_ECDSA_R1, BBOP_SCHNORR, BBOP_PEDERSEN, BBOP_FIXED_BASE };
typedef struct {
    uint32_t op;
    uint64_t a, b, c;        /* destination/condition, lhs/source, rhs */
    uint32_t sub_op, bit_size;
    uint64_t location;
    fr_t value;
    char *function; reg_or_mem_t *dests; size_t n_dests; reg_or_mem_t *inputs; size_t n_inputs;
    /* black box op: operands in declaration order; heap vector = (pointer,size-reg), heap array = (pointer,size) */
    uint32_t bbop; uint64_t bb[10];
} brillig_op_t;

typedef struct { int is_array; fr_t single; fr_t *arr; size_t n; } fc_output_t;
typedef struct { fc_output_t *values; size_t n; } fc_result_t;
typedef struct { int is_array; expr_t single; expr_t *arr; size_t n; } brillig_input_t;
typedef struct { int is_array; uint32_t w; uint32_t *arr; size_t n; } brillig_output_t;
typedef struct {
    brillig_input_t *inputs; size_t n_inputs;
    brillig_output_t *outputs; size_t n_outputs;
    fc_result_t *fc_results; size_t n_fc_results, cap_fc_results;
    brillig_op_t *bytecode; size_t n_bytecode;
    int has_predicate; expr_t predicate;
} brillig_t;

enum { OP_ARITHMETIC = 0, OP_BLACKBOX = 1, OP_DIRECTIVE = 2, OP_BRILLIG = 3, OP_MEMORY_OP = 4, OP_MEMORY_INIT = 5 };
typedef struct {
    uint32_t kind;
    expr_t expr;            /* arithmetic */
    bb_call_t bb;
    directive_t dir;
    brillig_t brillig;
    uint32_t block_id;
    expr_t mem_operation, mem_index, mem_value; int has_predicate; expr_t predicate; /* MemoryOp */
    uint32_t *init; size_t n_init; /* MemoryInit */
} opcode_t;

typedef struct { int is_brillig; uint64_t acir_index, brillig_index; char *message; } assert_msg_t;

typedef struct {
    uint32_t current_witness_index;
    opcode_t *opcodes; size_t n_opcodes;
    uint32_t *private_parameters; size_t n_private;
    uint32_t *public_parameters; size_t n_public;
    uint32_t *return_values; size_t n_return;
    assert_msg_t *assert_messages; size_t n_assert;
    uint32_t max_witness; /* highest witness index referenced anywhere */
} circuit_t;

/* Parse the *inflated* bincode bytes of a Circuit. Returns NULL on malformed input. */
circuit_t *acir_circuit_parse(const uint8_t *buf, size_t len);
void acir_circuit_free(circuit_t *c);
void expr_free(expr_t *e);
void expr_clone(expr_t *dst, const expr_t *src);

#ifdef __cplusplus
}
#endif
#endif
