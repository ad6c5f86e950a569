/*
 * oracle/pwg.h -- TEST INFRASTRUCTURE ONLY (CPU oracle). Not part of the shipped product path.
 *
 * CPU restatement of acvm::pwg (partial witness generation):
 *   /root/reference/acvm/src/pwg/mod.rs:33-51,72-114,129-372   ACVM, ACVMStatus, errors, helpers
 *   /root/reference/acvm/src/pwg/arithmetic.rs:27-239          ArithmeticSolver
 *   /root/reference/acvm/src/pwg/blackbox/                       dispatcher + range/logic/hash/...
 *   /root/reference/acvm/src/pwg/directives/mod.rs:23-122      Quotient / ToLeRadix
 *   /root/reference/acvm/src/pwg/memory_op.rs:16-124           MemoryOpSolver
 *   /root/reference/acvm/src/pwg/brillig.rs:20-150             BrilligSolver (brillig_vm.c)
 *   /root/reference/blackbox_solver/src/lib.rs:27-45           BlackBoxFunctionSolver trait (backend_t)
 * Numeric status / error codes are the same as include/acvm_amd.h so parity tests compare them directly.
 */
#ifndef ORACLE_PWG_H
#define ORACLE_PWG_H
#include "acir.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ACVMStatus (pwg/mod.rs:33-51) */
enum { ST_SOLVED = 0, ST_IN_PROGRESS = 1, ST_FAILURE = 2, ST_REQUIRES_FOREIGN_CALL = 3 };
/* OpcodeResolutionError (pwg/mod.rs:100-114) + OpcodeNotSolvable (:72-78) flattened */
enum {
    E_NONE = 0,
    E_MISSING_ASSIGNMENT = 1,   /* aux0 = witness index */
    E_TOO_MANY_UNKNOWNS = 2,    /* ExpressionHasTooManyUnknowns */
    E_UNSUPPORTED_BLACKBOX = 3, /* aux0 = BlackBoxFunc tag */
    E_UNSATISFIED = 4,          /* opcode_location = Resolved(Acir(opcode_index)) */
    E_INDEX_OOB = 5,            /* aux0 = index, aux1 = array_size */
    E_BLACKBOX_FAILED = 6,      /* aux0 = BlackBoxFunc tag, message */
    E_BRILLIG_FAILED = 7,       /* message, call_stack */
    E_PANIC = 8                 /* the reference would panic here (message) */
};

typedef struct {
    uint32_t status, err, opcode_index, aux0, aux1;
    uint32_t n_call_stack;
    uint32_t call_stack[16];
    char message[200];
} acvm_result_t;

/* BlackBoxFunctionSolver (blackbox_solver/src/lib.rs:27-45). Return: 0 ok, 1 Failed(err), 2 Unsupported, 3 panic */
typedef struct {
    void *ctx;
    int (*schnorr_verify)(void *ctx, const fr_t *pkx, const fr_t *pky, const uint8_t *sig, size_t sig_len,
                          const uint8_t *msg, size_t msg_len, int *ok, char *err, size_t errlen);
    int (*pedersen)(void *ctx, const fr_t *inputs, size_t n, uint32_t domain_separator, fr_t *x, fr_t *y,
                    char *err, size_t errlen);
    int (*fixed_base_scalar_mul)(void *ctx, const fr_t *low, const fr_t *high, fr_t *x, fr_t *y, char *err,
                                 size_t errlen);
} backend_t;

/* built-in backends: 0 = barretenberg restatement (grumpkin.c), 1 = StubbedBackend (acvm/tests/solver.rs:20-46,
 * panics when hit), 2 = DummyBlackBoxSolver (brillig_vm/src/lib.rs:392-420: true, (2,3), (4,5)) */
const backend_t *oracle_backend(int which);

typedef struct mem_block {
    uint32_t id, len, cap;
    fr_t *cells;
    uint8_t *present;
} mem_block_t;

typedef struct {
    char *function;
    fr_t **inputs; size_t *input_len; size_t n_inputs;
} foreign_call_wait_t;

/* btree.c: ordered u32 set shaped like Rust's BTreeMap nodes (the sparse-map cost model of the faithful CPU baseline) */
typedef struct oracle_btree oracle_btree_t;
oracle_btree_t *oracle_btree_new(void);
void oracle_btree_free(oracle_btree_t *t);
int oracle_btree_contains(const oracle_btree_t *t, uint32_t k);
int oracle_btree_insert(oracle_btree_t *t, uint32_t k);
size_t oracle_btree_len(const oracle_btree_t *t);

/* timing modes of the CPU baseline; results are bit-identical in every mode:
 *   ORACLE_MODE_SPARSE_MAP  every witness lookup / insert also descends a BTreeMap-shaped tree (cpu_ref_faithful)
 *   ORACLE_MODE_CACHE_INV   -1/coeff of an Arithmetic opcode's constant divisor is computed once per circuit and host thread
 *                           instead of once per solved witness (cpu_ref_dense: "best reasonable CPU") */
enum { ORACLE_MODE_SPARSE_MAP = 1, ORACLE_MODE_CACHE_INV = 2 };
typedef struct {
    fr_t *coeff, *inv; /* per opcode: the divisor last seen and its inverse */
    uint8_t *have;
} oracle_inv_cache_t;

typedef struct {
    const circuit_t *c;
    const backend_t *backend;
    int mode;
    oracle_btree_t *index;          /* ORACLE_MODE_SPARSE_MAP: the assigned witness ids */
    oracle_inv_cache_t *inv_cache;  /* ORACLE_MODE_CACHE_INV: shared by the instances of one host thread (not owned) */
    uint32_t nw;
    fr_t *val;
    uint8_t *assigned;
    size_t ip;
    acvm_result_t res;
    mem_block_t *blocks; size_t n_blocks;
    /* foreign call results appended by resolve_pending_foreign_call, per opcode (pwg/mod.rs:214-228) */
    fc_result_t **extra_fc; size_t *n_extra_fc;
    foreign_call_wait_t pending;
} oracle_acvm_t;

/* ACVM::new (pwg/mod.rs:146-156). values_be32: n_initial x 32 bytes big-endian, reduced mod p. */
oracle_acvm_t *oracle_acvm_new(const circuit_t *c, const backend_t *backend, size_t n_initial,
                               const uint32_t *ids, const uint8_t *values_be32);
/* same with a timing mode; cache may be NULL unless ORACLE_MODE_CACHE_INV is set */
oracle_acvm_t *oracle_acvm_new_mode(const circuit_t *c, const backend_t *backend, size_t n_initial, const uint32_t *ids,
                                    const uint8_t *values_be32, int mode, oracle_inv_cache_t *cache);
oracle_inv_cache_t *oracle_inv_cache_new(const circuit_t *c);
void oracle_inv_cache_free(oracle_inv_cache_t *k);
void oracle_acvm_free(oracle_acvm_t *a);
/* ACVM::new again on the same object for the next instance of the same circuit (same ids): buffers are kept */
void oracle_acvm_reset(oracle_acvm_t *a, size_t n_initial, const uint32_t *ids, const uint8_t *values_be32);
/* ACVM::solve (pwg/mod.rs:236-241), ACVM::solve_opcode (:243-303) */
uint32_t oracle_acvm_solve(oracle_acvm_t *a);
uint32_t oracle_acvm_solve_opcode(oracle_acvm_t *a);
/* ACVM::resolve_pending_foreign_call (:214-228): takes ownership of nothing, copies values */
int oracle_acvm_resolve_foreign_call(oracle_acvm_t *a, const fc_result_t *result);

/* shared helpers (pwg/mod.rs:309-372, arithmetic.rs:212-239) used by brillig_vm.c */
int pwg_get_value(oracle_acvm_t *a, const expr_t *e, fr_t *out);       /* 0 ok, else sets a->res error */
int pwg_insert_value(oracle_acvm_t *a, uint32_t w, const fr_t *v);      /* 0 ok, else E_UNSATISFIED */
void pwg_fail(oracle_acvm_t *a, uint32_t err, uint32_t aux0, uint32_t aux1, const char *msg);
/* ecdsa.c: 0 / 1, or a negative panic code (see oracle_ecdsa_panic_text); curve 0 = secp256k1, 1 = secp256r1 */
int oracle_ecdsa_verify(int curve, const uint8_t *hashed_msg, size_t msg_len, const uint8_t pkx[32], const uint8_t pky[32], const uint8_t sig[64]);
const char *oracle_ecdsa_panic_text(int code);
/* sorting.c: Directive::PermutationSort */
int oracle_solve_permutation_sort(oracle_acvm_t *a, const directive_t *d);
size_t oracle_sorting_route(const uint32_t *inputs, const uint32_t *outputs, uint32_t n, uint8_t *bits);
/* brillig_vm.c */
int brillig_solve(oracle_acvm_t *a, const brillig_t *b, size_t acir_index); /* 0 ok, 1 err (res set), 2 foreign call wait */

#ifdef __cplusplus
}
#endif
#endif
