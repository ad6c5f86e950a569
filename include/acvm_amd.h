/*
 * acvm_amd.h -- C ABI of the MI355X-native batched ACIR witness solver (libacvm_amd.so).
 *
 * This header is the drop-in boundary for ONE hot path of noir-lang/acvm v0.27.0:
 * `acvm::pwg::ACVM::solve()` and what it calls, for B independent witness instances of one circuit.
 * Plain pointers and sizes only; no torch / HIP types. Every entry point cites the reference
 * interface it replaces (paths relative to the acvm repository root).
 *
 *   reference                                               this ABI
 *   ------------------------------------------------------  -------------------------------------------
 *   acir::circuit::Circuit::read   circuit/mod.rs:154-161    acvm_circuit_from_bytes
 *   ACVM::new(backend, opcodes, initial_witness)
 *                         acvm/src/pwg/mod.rs:146-156        acvm_batch_new + acvm_batch_set_initial_witness
 *   ACVM::solve           acvm/src/pwg/mod.rs:236-241        acvm_batch_solve
 *   ACVMStatus / OpcodeResolutionError  mod.rs:33-51,100-114 acvm_result_t via acvm_batch_results
 *   ACVM::witness_map / finalize        mod.rs:161,176-181   acvm_batch_witness_map / acvm_batch_witness
 *   ACVM::instruction_pointer           mod.rs:171           acvm_result_t.opcode_index
 *   ACVM::get_pending_foreign_call      mod.rs:203-209       acvm_batch_pending_foreign_call*
 *   ACVM::resolve_pending_foreign_call  mod.rs:214-228       acvm_batch_resolve_foreign_call
 *   trait BlackBoxFunctionSolver  blackbox_solver/src/lib.rs:27-45   acvm_bb_solver_t (vtable)
 *
 * Threading (reference: `&mut self`, backend not Sync): one batch handle = one host thread + one HIP
 * device + one stream set. Handles are independent, also on one device; the only process-wide state is read-mostly
 * (the tuning table; the per-device lookup tables, built once under a per-device lock and reference-counted by the handles).
 * acvm_node_* drives one handle per device.
 * Errors: functions return 0 on success or a negative ACVM_E_* code; acvm_last_error() gives text.
 * The library requires a gfx950 device for every compute entry point and fails loudly without one;
 * there is no CPU fallback.
 */
#ifndef ACVM_AMD_H
#define ACVM_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ACVM_AMD_ABI_VERSION 6

/* library-level error codes */
enum {
    ACVM_OK = 0,
    ACVM_E_INVALID = -1,     /* bad argument / handle */
    ACVM_E_MALFORMED = -2,   /* circuit bytes do not decode (the reference would panic in bincode::deserialize) */
    ACVM_E_UNSUPPORTED = -3, /* opcode outside the accelerated set -- refused at batch creation (an instance whose Brillig program
                                exceeds the device's VM limits is a per-instance outcome, ACVM_ERR_DEVICE_LIMIT, not a failed call) */
    ACVM_E_DEVICE = -4,      /* no gfx950 device / HIP runtime error */
    ACVM_E_STATE = -5,       /* call not valid in the current state (reference: panic) */
    ACVM_E_NOMEM = -6        /* host allocation failed (nothing unwinds through this ABI) */
};

/* ACVMStatus (acvm/src/pwg/mod.rs:33-51) */
enum { ACVM_STATUS_SOLVED = 0, ACVM_STATUS_IN_PROGRESS = 1, ACVM_STATUS_FAILURE = 2, ACVM_STATUS_REQUIRES_FOREIGN_CALL = 3 };

/* OpcodeResolutionError (mod.rs:100-114) with OpcodeNotSolvable (mod.rs:72-78) flattened */
enum {
    ACVM_ERR_NONE = 0,
    ACVM_ERR_MISSING_ASSIGNMENT = 1,   /* aux0 = witness index */
    ACVM_ERR_TOO_MANY_UNKNOWNS = 2,    /* OpcodeNotSolvable::ExpressionHasTooManyUnknowns */
    ACVM_ERR_UNSUPPORTED_BLACKBOX = 3, /* aux0 = BlackBoxFunc tag */
    ACVM_ERR_UNSATISFIED = 4,          /* UnsatisfiedConstrain{Resolved(Acir(opcode_index))} */
    ACVM_ERR_INDEX_OOB = 5,            /* IndexOutOfBounds{index = aux0, array_size = aux1} */
    ACVM_ERR_BLACKBOX_FAILED = 6,      /* BlackBoxFunctionFailed(func = aux0, message) */
    ACVM_ERR_BRILLIG_FAILED = 7,       /* BrilligFunctionFailed{message, call_stack} */
    ACVM_ERR_PANIC = 8,                /* the reference would panic at this opcode (message) */
    /* NOT a reference outcome. The reference's Brillig VM has no resource limits (brillig_vm/src/memory.rs:27-39, lib.rs:154-307); the device
     * runs a program with limits, raises them on the exact path, and past the stated maxima of the library (acvm_tuning_set: brillig_steps_max_log2,
     * brillig_call_depth_max, brillig_mem_max_log2) or of the device's memory THIS INSTANCE ends here: status Failure at its Brillig opcode,
     * aux0 = ACVM_LIMIT_* (what was reached), aux1 = the limit. It says "this library could not finish the instance", not "the circuit is
     * unsatisfied": re-run the instance with the reference (or with raised maxima). Every other instance of the batch keeps its result, as the
     * reference's caller loop loses one instance at most (acvm_js/src/execute.rs:60-119). */
    ACVM_ERR_DEVICE_LIMIT = 9
};
enum { ACVM_LIMIT_BRILLIG_STEPS = 1, ACVM_LIMIT_BRILLIG_CALL_DEPTH = 2, ACVM_LIMIT_BRILLIG_MEMORY = 3 /* cells of 32 bytes */,
       ACVM_LIMIT_DEVICE_MEMORY = 4 /* MiB of VM scratch the device could not provide */ };

/* Per-instance outcome. Same layout and numbering as the CPU oracle's result record. */
typedef struct {
    uint32_t status;       /* ACVM_STATUS_* */
    uint32_t err;          /* ACVM_ERR_* when status == FAILURE */
    uint32_t opcode_index; /* instruction pointer of the failing opcode (ACVM::instruction_pointer) */
    uint32_t aux0, aux1;
    uint32_t n_call_stack;
    uint32_t call_stack[16]; /* Brillig indices for BrilligFunctionFailed */
    char message[200];
} acvm_result_t;

/*
 * BlackBoxFunctionSolver (blackbox_solver/src/lib.rs:27-45) as a vtable. Field elements cross as 32-byte
 * canonical big-endian. Return 0 = Ok, 1 = BlackBoxResolutionError::Failed(err text), 2 = Unsupported, 3 = the
 * implementation panicked (err text; the instance fails with ACVM_ERR_PANIC -- what StubbedBackend does).
 * A NULL solver selects the built-in HIP implementation of barretenberg's three functions.
 *
 * The *_batch members are optional (NULL = call the per-instance function n times): one call serves the n instances of a
 * batch that reach the opcode. Arrays are instance-major host memory; rc[i] uses the return codes above, err is
 * [n][err_stride] NUL-terminated texts. They are what a backend that is itself batched (a GPU backend, a thread pool)
 * implements; the library gathers all instances once, makes ONE call per opcode and scatters the results once.
 *
 * The solver is the Brillig VM's solver too, as in the reference (brillig_vm/src/lib.rs:61,81,298; black_box.rs:139-163): a Brillig
 * program's BlackBoxOp::{SchnorrVerify, Pedersen, FixedBaseScalarMul} reach the same table (ABI 5; refused before). Inside
 * acvm_batch_solve / acvm_batch_solve_opcode every instance that stands at such an op is answered in one pass (one *_batch call per call
 * shape) and the opcode re-runs its VM with the answers; a callback's failure fails the VM at the op with BlackBoxResolutionError's Display
 * string. The caller never sees these round trips. ACVM_BATCH_REUSE_SLOTS takes a solver for the ACIR-level black boxes; a solver inside
 * Brillig is refused there with every other foreign call.
 */
typedef struct {
    void *ctx;
    int (*schnorr_verify)(void *ctx, const uint8_t pkx[32], const uint8_t pky[32], const uint8_t *sig, size_t sig_len,
                          const uint8_t *msg, size_t msg_len, uint8_t *ok, char *err, size_t err_len);
    int (*pedersen)(void *ctx, const uint8_t *inputs_be32, size_t n_inputs, uint32_t domain_separator, uint8_t x[32],
                    uint8_t y[32], char *err, size_t err_len);
    int (*fixed_base_scalar_mul)(void *ctx, const uint8_t low[32], const uint8_t high[32], uint8_t x[32], uint8_t y[32],
                                 char *err, size_t err_len);
    /* pk_be32 [n][2][32], sig [n][sig_len], msg [n][msg_len], ok [n] */
    int (*schnorr_verify_batch)(void *ctx, size_t n, const uint8_t *pk_be32, const uint8_t *sig, size_t sig_len, const uint8_t *msg,
                                size_t msg_len, uint8_t *ok, uint8_t *rc, char *err, size_t err_stride);
    /* inputs_be32 [n][n_inputs][32], xy_be32 [n][2][32] */
    int (*pedersen_batch)(void *ctx, size_t n, const uint8_t *inputs_be32, size_t n_inputs, uint32_t domain_separator, uint8_t *xy_be32,
                          uint8_t *rc, char *err, size_t err_stride);
    /* low_high_be32 [n][2][32], xy_be32 [n][2][32] */
    int (*fixed_base_scalar_mul_batch)(void *ctx, size_t n, const uint8_t *low_high_be32, uint8_t *xy_be32, uint8_t *rc, char *err,
                                       size_t err_stride);
} acvm_bb_solver_t;

/* The two fakes of the reference's own tests as ready-made vtables (static storage, never freed):
 *   acvm_bb_stubbed  StubbedBackend       acvm/tests/solver.rs:20-46      every function panics "Path not trodden by this test"
 *   acvm_bb_dummy    DummyBlackBoxSolver  brillig_vm/src/lib.rs:392-420   schnorr_verify = true, pedersen = (2, 3), fixed_base = (4, 5) */
const acvm_bb_solver_t *acvm_bb_stubbed(void);
const acvm_bb_solver_t *acvm_bb_dummy(void);

typedef struct acvm_circuit acvm_circuit_t;
typedef struct acvm_batch acvm_batch_t;

/* Statistics of the static plan and of the last solve (measurement). */
typedef struct {
    uint32_t n_opcodes, n_witnesses, n_levels, n_fast_gates, n_dyn_gates;
    uint32_t max_level_width, n_kernel_launches, n_slow_instances;
    uint64_t algorithmic_bytes_per_instance; /* sum over gates of 32 B x (distinct known operands + written witness) */
    uint64_t arith_algorithmic_bytes_per_instance; /* the part moved by arith_level_kernel */
    double plan_ms;         /* one-time levelisation, host */
    double solve_device_ms; /* HIP events around the whole last solve, on the batch's stream */
    double arith_kernel_ms; /* sum of HIP-event durations of the arithmetic level kernels of the last solve */
    double slow_path_ms;
    double dyn_kernel_ms;   /* same for the batched denominator inversions (inverse_batch_kernel, beside the former on a 2nd stream) */
    uint64_t dyn_algorithmic_bytes_per_instance;
    /* non-arithmetic opcodes, by kernel class: 0 light (range / logic / directives / memory), 1 hashes, 2 Grumpkin, 3 Brillig */
    uint32_t n_other_records;
    uint32_t truncated_at; /* first opcode the generic instance cannot execute (all instances take the exact kernels from there), or 0xFFFFFFFF */
    uint64_t class_algorithmic_bytes_per_instance[4];
    double class_kernel_ms[4]; /* summed HIP-event durations of the class's level kernels of the last solve (profiling on) */
    uint32_t n_gate_pairs;     /* arithmetic gates that run in their producer's wave and take its output from registers */
    uint32_t n_inverse_slots;  /* rows of the inverse table (denominators of the n_dyn_gates gates, rows reused) */
    uint32_t n_scaled_witnesses; /* witnesses the level kernels keep as scale x value (unscaled on export and for the exact path) */
    uint32_t n_arith_launches; /* launches of arith_level_kernel per solve (levels that hold gates) */
    uint32_t n_table_rows;     /* rows of the device witness table: n_witnesses, or fewer with ACVM_BATCH_REUSE_SLOTS */
    uint32_t n_digest_segments; /* records of leaves of the folded digest (ACVM_BATCH_FOLD_DIGEST), 0 if the digest is not folded */
    uint32_t n_brillig_inlined; /* Brillig opcodes whose straight-line program the level schedule runs as a light record (no VM) */
    uint32_t n_brillig_retries; /* passes of the last solve that re-ran Brillig opcodes of the exact path with raised VM limits */
    uint32_t n_hash_chained;   /* byte-message hashes that run in the workgroup of the hash whose digest they consume (no launch of their own) */
    /* relaxed rows (ABI 5): SOLVE gates that store their result as the column scan left it (any representative below 2^256) / after one
     * quotient-estimate reduction (below 1.03 p) / canonical; the largest bound a gate's result reaches, in units of p / 256 */
    uint32_t n_gate_out_asis, n_gate_out_weak, n_gate_out_canon, max_gate_bound;
    /* byte planes (ABI 5): initial witnesses that byte-message hashes read carry a 4-byte copy per instance (low limb + is-byte flag) written by the
     * import; n_byte_plane_reads inputs of hash records read it in place of the 32-byte row. The algorithmic-byte figures above keep the reference's
     * unit (32 bytes per witness read): what the hash kernel itself moves is 28 bytes less per such input, what the import moves 4 bytes more per plane. */
    uint32_t n_byte_planes, n_byte_plane_reads;
    /* ABI 6: launches of the level schedule by stream (main, inversions, the three heavy lanes, the digest lane) and the cross-stream waits of one solve */
    uint32_t n_stream_launches[6], n_stream_waits;
} acvm_stats_t;

const char *acvm_last_error(void);
int acvm_abi_version(void);
int acvm_device_count(void);
int acvm_set_device(int device);
int acvm_device_synchronize(void);
/* name of the current device's gcnArch ("gfx950...") into out */
int acvm_device_arch(char *out, size_t out_len);
/*
 * The per-device lookup tables (Grumpkin: 3 MB of fixed-base and Pedersen slice tables, 268 MB of 16-bit windows, the 503 MB pair table
 * or the 23.6 GB window table of the level Pedersen kernel; ECDSA: 2 x 64 MiB of generator windows) are built on the device at the first
 * handle whose circuit needs them -- under a lock of that device only, on a stream of their own -- and shared by every later handle of the
 * device. They stay until this call: frees the tables of `device` and returns the bytes given back, or ACVM_E_STATE while a handle of
 * that device still uses them (free the handles first). The next handle that needs a table rebuilds it (0.3 s for the largest).
 * acvm_tuning_set("tables_keep", 0) makes the destruction of a device's last handle release them without this call.
 * The reference keeps its counterpart -- the wasm instance with barretenberg's tables -- for the lifetime of the solver object
 * (barretenberg_blackbox_solver/src/wasm/mod.rs:59-82).
 */
long long acvm_device_release_tables(int device);

/*
 * Planner / scheduler modes and the device's Brillig VM limits, process-wide (csrc/tuning.hpp lists every key with its default;
 * acvm_tuning_key(i) enumerates them, NULL past the end). A batch keeps the values it was created with. No mode changes a result --
 * the parity tests sweep them against the oracle; the defaults are the measured optimum. The limits: the reference's Brillig VM has
 * none (brillig_vm/src/memory.rs:27-39 grows memory on write, lib.rs:154-307 runs any number of steps at any call depth); the device
 * runs with brillig_steps_log2 / brillig_call_depth / the planner's memory estimate, retries an instance that reaches one with the
 * limit raised, and past brillig_steps_max_log2 / brillig_call_depth_max / brillig_mem_max_log2 that instance ends with
 * ACVM_ERR_DEVICE_LIMIT (the others keep their results). Command-line tools may preset values through the environment: ACVM_TUNING="key=value,key=value".
 */
int acvm_tuning_set(const char *key, long long value);
int acvm_tuning_get(const char *key, long long *value);
const char *acvm_tuning_key(unsigned index);

/* Device self test of the field library: n pseudo-random operand pairs; returns the number of lanes whose
 * hand-scheduled routines disagree with the portable ones (0 = pass), or a negative error. */
int acvm_selftest(uint32_t n, uint64_t seed);

/* Component probes of the Grumpkin kernels (parity tests of barretenberg's building blocks against SURVEY Appendix A):
 * what 0 = host table point (param = table << 24 | index; tables: 0 Pedersen k*D[i] at i*512+k-1, 1 8-bit windows,
 * 2 ladder k*D[3j+1], 3 skew D[3j+2]); 1 = device hash_single(in[0], parity param); 2 = device hash-ladder
 * compress(in[0..n_in)), 3 = device fixed-base product (window table param, 256-bit integer in[0]); 4 = device table point;
 * 8 / 9 = in[0] * (in[1], in[2]) for a 256-bit integer and an affine point: 8 through SchnorrVerify's GLV + window-table path, 9 by double-and-add.
 * in: n_in x 32 bytes big-endian; out: 64 bytes (x || y) big-endian. */
int acvm_debug_grumpkin(uint32_t what, uint32_t param, const uint8_t *in_be32, uint32_t n_in, uint8_t *out_be64);
/* Component probes of the ECDSA kernels (secp_device.hpp run ON THE DEVICE against integers: k256 / p256 are the spec of blackbox_solver/src/lib.rs:66-210,
 * the tests compare with Python): curve 0 = secp256k1, 1 = secp256r1; one device lane per item. what: 0 a b -> a b; 1 a -> a^2; 2 a b -> a + b; 3 a b -> a - b;
 * 4 a -> 1 / a (0 for 0); 5 a -> a^((p + 1) / 4), all mod p; 6 X Y Z -> the Jacobian double; 7 X Y Z x y -> the Jacobian sum with the affine point (x, y);
 * compositions the formulas are made of: 8 a -> a^4 (a product fed to a product); 9 a b -> (a + b)^2 (an unreduced sum into a product); 10 a b -> a - 4 b
 * (a difference against a multiple of p); 11 a -> a^32 (the squaring loop of the square-root chains).
 * in: n_items x (2, 1, 2, 2, 1, 1, 3, 5, 1, 2, 2, 1) x 32 bytes big-endian, values < p; out: n_items x (1, 1, 1, 1, 1, 1, 3, 3, 1, 1, 1, 1) x 32 bytes. */
int acvm_debug_secp(uint32_t curve, uint32_t what, const uint8_t *in_be32, uint32_t n_items, uint8_t *out_be32);

/* Peak of the ALU roofline of the integer-bound kernels: back-to-back Montgomery products (fr29_mul, the product every
 * kernel uses) on every SIMD, waves_per_simd dependent chains of 2 * iters products interleaved per SIMD; the best of three timed
 * launches as modmul/s, and the number of products one launch executes. */
int acvm_debug_modmul_rate(uint32_t iters, uint32_t waves_per_simd, double *modmul_per_s, uint64_t *n_modmul);
/* The same for the ECDSA kernels: back-to-back base-field products of secp256k1 (curve 0) / secp256r1 (curve 1), a product and a square in turn
 * (acvm_amd/csrc/secp_device.hpp sp_mul / sp_sqr). */
int acvm_debug_secp_rate(uint32_t curve, uint32_t iters, uint32_t waves_per_simd, double *products_per_s, uint64_t *n_products);
/* The measured streaming ceiling of the device for the gate kernel's access shape (reported beside the 8 TB/s spec peak of the HBM
 * roofline): two rows of `bytes` bytes read and one written, 16 bytes per lane, the grid covering the data like a level launch;
 * bytes moved (3 x bytes) per second of the best of four launches, in GB/s. */
int acvm_debug_stream_rate(size_t bytes, double *gb_per_s);

/* Circuit::read: gzip(bincode) or raw bincode bytes. */
acvm_circuit_t *acvm_circuit_from_bytes(const uint8_t *bytes, size_t len);
void acvm_circuit_free(acvm_circuit_t *c);
uint32_t acvm_circuit_num_opcodes(const acvm_circuit_t *c);
uint32_t acvm_circuit_num_witnesses(const acvm_circuit_t *c); /* 1 + highest witness index referenced */
/*
 * ACVM::opcodes (acvm/src/pwg/mod.rs:166-168) seen through the ABI: for opcodes [first, first + n) of the circuit, kinds[2 i] = the variant of
 * acir::circuit::Opcode in declaration order (opcodes.rs:15-34: 0 Arithmetic, 1 BlackBoxFuncCall, 2 Directive, 3 Brillig, 4 MemoryOp,
 * 5 MemoryInit) and kinds[2 i + 1] = what tells its instances apart at a glance: the BlackBoxFuncCall variant (black_box_function_call.rs:20-115,
 * the tags of ACVM_ERR_UNSUPPORTED_BLACKBOX's aux0), the Directive variant (0 Quotient, 1 ToLeRadix, 2 PermutationSort), the length of a
 * Brillig bytecode, the block id of a memory opcode, 0 for Arithmetic. With acvm_circuit_num_opcodes and acvm_result_t.opcode_index (the
 * instruction pointer) a binding serves `opcodes()` from the Vec<Opcode> it was constructed with and can assert it is the circuit the
 * handle runs.
 */
int acvm_circuit_opcode_kinds(const acvm_circuit_t *c, uint32_t first, uint32_t n, uint32_t *kinds /*[n][2]*/);
/* Host-only levelisation against a set of initial witness ids (no device needed): plan statistics in *out. Returns 0, or
 * ACVM_E_UNSUPPORTED (reason in acvm_last_error) if the circuit holds an opcode no kernel implements. */
int acvm_circuit_plan_stats(const acvm_circuit_t *c, const uint32_t *initial_ids, uint32_t n_initial, acvm_stats_t *out);
/* how often this circuit handle has been levelised so far: handles created for the same (initial ids, options, tuning) share one immutable plan */
uint64_t acvm_circuit_plans_built(const acvm_circuit_t *c);
/* the same for a batch created with acvm_batch_new_ex's flags (n_table_rows, n_digest_segments) */
int acvm_circuit_plan_stats_ex(const acvm_circuit_t *c, const uint32_t *initial_ids, uint32_t n_initial, uint32_t flags, const uint32_t *keep_ids,
                               uint32_t n_keep, acvm_stats_t *out);

/*
 * Host-only: the hazard checker of the level schedule (acvm_amd/csrc/schedule_check.cpp). One solve is enqueued on up to six streams with
 * partial waits, recycled rows and rows whose representation depends on the consumer; the reference's semantics are strictly in order
 * (acvm/src/pwg/mod.rs:236-303). For the plan and schedule a handle of n_instances instances with these flags (acvm_batch_new_ex's, | 0x100 =
 * planned for a caller-supplied solver) would use, the checker derives every launch's reads and writes from the record words, builds
 * happens-before from stream order + event edges, and proves: every conflicting pair of accesses is ordered, every read sees the write of the
 * witness the original opcode names, every reader outside the gate kernels sees a canonical row, no bound passes 2^256. Returns 0 (proved),
 * 1 (findings; the first 32 as text in report) or a negative error. counts[0..5) = launches, waits, accesses, records, findings.
 * drop_wait = k leaves the k-th cross-stream wait out of the schedule (mutation testing: the checker must then name the resource and the
 * two launches); 0xFFFFFFFF = the schedule as enqueued.
 */
int acvm_circuit_check_schedule(const acvm_circuit_t *c, const uint32_t *initial_ids, uint32_t n_initial, uint32_t flags, const uint32_t *keep_ids, uint32_t n_keep,
                                uint32_t n_instances, uint32_t drop_wait, uint64_t *counts /*[5]*/, char *report, size_t report_len);
/* Host-only: 64-bit fingerprints of the static plan (gate stream, in-order program, level and dependency tables, row assignment ...), one per
 * component, into out[0, cap); returns the number of components. flags: acvm_batch_new_ex's, | 0x100 = planned for a caller-supplied solver.
 * For tests and tools that assert that a change of the planner moved nothing (tests/test_plan_host.py). */
int acvm_debug_plan_fingerprint(const acvm_circuit_t *c, const uint32_t *initial_ids, uint32_t n_initial, uint32_t flags, const uint32_t *keep_ids,
                                uint32_t n_keep, uint64_t *out, uint32_t cap);
/*
 * ACVM::new for n_instances instances that all assign the same initial witness ids.
 * The circuit is levelised once against that set. `solver` may be NULL (built-in HIP backend).
 */
acvm_batch_t *acvm_batch_new(const acvm_circuit_t *c, const acvm_bb_solver_t *solver, uint32_t n_instances,
                             const uint32_t *initial_ids, uint32_t n_initial);
/*
 * The same with options for callers that keep only the return witnesses and a digest of every map:
 *   ACVM_BATCH_FOLD_DIGEST  the per-instance digest (acvm_batch_digest) is computed DURING the solve, leaf by leaf as the witnesses
 *                           of a segment complete, beside the level kernels: acvm_batch_digest then only hashes the leaves.
 *   ACVM_BATCH_REUSE_SLOTS  witness-slot liveness reuse: a witness occupies a row of the device table from the level that writes it
 *                           to the level of its last reader, rows are recycled (implies the folded digest: a row is hashed before
 *                           it is reused). Afterwards only the initial witnesses and keep_ids can be read back (acvm_batch_witness
 *                           / _extract_witnesses), plus results and digests; acvm_batch_witness_map returns ACVM_E_STATE. Instances
 *                           that leave the generic path are re-solved from their initial witnesses in a table of their own
 *                           (all witnesses x flagged instances, padded to 64; acvm_batch_solve returns ACVM_E_UNSUPPORTED when
 *                           that table would be larger than both the level table and 8 GiB: solve such a tile without the flag).
 */
#define ACVM_BATCH_FOLD_DIGEST 1u
#define ACVM_BATCH_REUSE_SLOTS 2u
acvm_batch_t *acvm_batch_new_ex(const acvm_circuit_t *c, const acvm_bb_solver_t *solver, uint32_t n_instances, const uint32_t *initial_ids,
                                uint32_t n_initial, uint32_t flags, const uint32_t *keep_ids, uint32_t n_keep);
void acvm_batch_free(acvm_batch_t *b);
/* values_be32: [n_instances][n_initial][32] canonical big-endian (reduced mod p like from_be_bytes_reduce). */
int acvm_batch_set_initial_witness(acvm_batch_t *b, const uint8_t *values_be32);
/* same, from a device-resident buffer of the same layout (no PCIe in the call) */
int acvm_batch_set_initial_witness_device(acvm_batch_t *b, const void *d_values_be32);
/* Device buffers for callers that keep their inputs resident (acvm_batch_set_initial_witness_device): plain hipMalloc / hipFree /
 * hipMemcpy on the current device, so that a caller needs no HIP headers. */
void *acvm_device_malloc(size_t bytes);
int acvm_device_free(void *p);
int acvm_device_upload(void *dst_device, const void *src_host, size_t bytes);
/* ACVM::solve for every instance. Returns the number of instances not Solved, or a negative error. */
int acvm_batch_solve(acvm_batch_t *b);
/*
 * The same solve for a caller that runs tile after tile through one handle (a 10k-gate circuit at 2^20 instances is eight tiles of one
 * handle's table): d_next_values_be32 is the device buffer of the NEXT tile's initial witnesses (layout of acvm_batch_set_initial_witness_device,
 * valid until the solve after this one returns). Its import is enqueued behind this solve and gated on the device: it runs if, and only if, no
 * instance of this solve left the generic path (the exact path needs this tile's rows otherwise). The call returns when this solve's outcome
 * is known, while that import may still be running; the following acvm_batch_set_initial_witness_device with the SAME pointer then costs
 * nothing (or performs the import, had it been held back), and the next tile's kernels follow the import without a gap. Until then results,
 * statistics and every witness that is not an initial one can be read as after acvm_batch_solve; reading an initial witness, a whole map
 * or an unfolded digest returns ACVM_E_STATE -- those rows may already hold the next tile. Not with a caller-supplied solver, stepping or
 * resumed foreign calls (then it is acvm_batch_solve). The reference has no counterpart: one ACVM solves one instance (pwg/mod.rs:236-241).
 */
int acvm_batch_solve_then_import(acvm_batch_t *b, const void *d_next_values_be32);
/*
 * ACVM::solve_opcode (acvm/src/pwg/mod.rs:243-303) for the batch: executes ONE opcode -- the one at the smallest instruction
 * pointer among the instances that are InProgress -- for every InProgress instance standing on it, through the exact in-order
 * kernels (one lane per instance). Instances only ever differ in their instruction pointer after a foreign call: one that
 * waited is behind the others once it is resolved, and catches up one call at a time. Starts from a batch whose initial
 * witness was just set (or acvm_batch_reset); ACVM_E_STATE after a plain acvm_batch_solve. acvm_batch_solve after some steps
 * runs the rest. Returns the number of instances not Solved; acvm_batch_results gives status and instruction pointer
 * (InProgress instances report the opcode they stand on).
 */
int acvm_batch_solve_opcode(acvm_batch_t *b);
/* back to the state right after set_initial_witness (same inputs, nothing solved) */
int acvm_batch_reset(acvm_batch_t *b);
/* The handle serves any batch size up to the n_instances it was created with (its plan and tables are kept): from the next
 * acvm_batch_set_initial_witness on, every call covers instances [0, n), 1 <= n <= that capacity; the lanes behind n are dead. A caller whose
 * batches vary in size creates ONE handle for the largest (planning a 10^6-opcode circuit takes seconds) instead of one per size. */
int acvm_batch_set_instances(acvm_batch_t *b, uint32_t n);
/* force every instance through the exact in-order kernel instead of the level-parallel one (validation) */
int acvm_batch_set_force_slow_path(acvm_batch_t *b, int on);
int acvm_batch_results(acvm_batch_t *b, acvm_result_t *out /*[n_instances]*/);
/* one witness across the batch: out_be32 [n_instances][32], assigned [n_instances] (0/1) */
int acvm_batch_witness(acvm_batch_t *b, uint32_t witness, uint8_t *out_be32, uint8_t *assigned);
/* full witness maps of instances [first, first+n): assigned [n][nw], values_be32 [n][nw][32] (zeros if unassigned) */
int acvm_batch_witness_map(acvm_batch_t *b, uint32_t first, uint32_t n, uint8_t *assigned, uint8_t *values_be32);
int acvm_batch_stats(acvm_batch_t *b, acvm_stats_t *out);

/*
 * Brillig foreign calls. An instance whose Brillig VM reaches a ForeignCall without a result stops with status
 * ACVM_STATUS_REQUIRES_FOREIGN_CALL and its instruction pointer on that opcode (pwg/mod.rs:262-268).
 *   acvm_batch_pending_foreign_call        = ACVM::get_pending_foreign_call (mod.rs:203-209): returns 1 and fills *info if
 *                                            the instance waits, 0 if it does not
 *   acvm_batch_pending_foreign_call_inputs = ForeignCallWaitInfo::inputs (pwg/brillig.rs:157-163): lens[n_inputs], then the
 *                                            values of all inputs back to back, 32 bytes big-endian each (info.n_values)
 *   acvm_batch_resolve_foreign_call        = ACVM::resolve_pending_foreign_call (mod.rs:214-228): one ForeignCallResult =
 *                                            n_values outputs, output i a single value (is_array[i] == 0) or an array of
 *                                            lens[i] values; values back to back. ACVM_E_STATE if the instance is not waiting
 *                                            (the reference panics).
 * After resolving any number of instances, acvm_batch_solve continues them (the opcode re-runs its VM with the results so far).
 */
typedef struct {
    uint32_t opcode_index;   /* the Brillig opcode the instance waits at */
    uint32_t brillig_index;  /* index of the ForeignCall inside its bytecode */
    uint32_t n_inputs, n_values;
    char function[64];
} acvm_foreign_call_info_t;
int acvm_batch_pending_foreign_call(acvm_batch_t *b, uint32_t instance, acvm_foreign_call_info_t *info);
int acvm_batch_pending_foreign_call_inputs(acvm_batch_t *b, uint32_t instance, uint32_t *lens, uint8_t *values_be32);
int acvm_batch_resolve_foreign_call(acvm_batch_t *b, uint32_t instance, uint32_t n_values, const uint8_t *is_array, const uint32_t *lens,
                                    const uint8_t *values_be32);
/* enable per-kernel HIP-event timing of the level kernels (small overhead) */
int acvm_batch_set_profiling(acvm_batch_t *b, int on);

/*
 * What a caller does right after solve.
 *
 * Circuit::get_assert_message (acir/src/circuit/mod.rs:43-51) for OpcodeLocation::Acir(acir_index) (brillig_index =
 * ACVM_LOCATION_ACIR) or OpcodeLocation::Brillig{acir_index, brillig_index}. Copies the NUL-terminated message (truncated to
 * cap) and returns its full length, or returns -1 when the location has no message.
 */
#define ACVM_LOCATION_ACIR 0xFFFFFFFFu
int acvm_circuit_assert_message(const acvm_circuit_t *c, uint32_t acir_index, uint32_t brillig_index, char *out, size_t cap);
/*
 * Witness sets of the circuit, ascending (they are BTreeSets in the reference): private_parameters, public_parameters,
 * return_values (circuit/mod.rs:25-32), public_inputs() = public_parameters U return_values (:115-121),
 * circuit_arguments() = private U public parameters (:109-113). Writes up to cap indices, returns the size of the set.
 */
enum { ACVM_SET_PRIVATE_PARAMETERS = 0, ACVM_SET_PUBLIC_PARAMETERS = 1, ACVM_SET_RETURN_VALUES = 2, ACVM_SET_PUBLIC_INPUTS = 3,
       ACVM_SET_CIRCUIT_ARGUMENTS = 4 };
int acvm_circuit_witness_set(const acvm_circuit_t *c, int which, uint32_t *out, uint32_t cap);
/*
 * The error string acvm_js reports for a failed instance (acvm_js/src/execute.rs:79-108): "Assertion failed: <message>" when
 * the failing location -- the opcode of UnsatisfiedConstrain / IndexOutOfBounds, the last call-stack entry of
 * BrilligFunctionFailed -- has an assert message in `c`, else the Display text of the OpcodeResolutionError
 * (acvm/src/pwg/mod.rs:100-114). Returns the full length; 0 and an empty string for an instance that did not fail.
 * ExpressionHasTooManyUnknowns carries its expression like the reference's Display: the opcode partially evaluated on the instance's map
 * for Opcode::Arithmetic (acvm/src/pwg/arithmetic.rs:31,38-42), the offending input expression for Opcode::Brillig (brillig.rs:46-74), in the
 * formats of acir_field/src/generic_ark.rs:13-74 and acir/src/circuit/opcodes.rs:88-102 (needs `c`; without it the text ends after "unknowns ").
 */
int acvm_batch_error_string(acvm_batch_t *b, const acvm_circuit_t *c, uint32_t instance, char *out, size_t cap);
/*
 * The same expression as DATA: OpcodeNotSolvable::ExpressionHasTooManyUnknowns(Expression) (acvm/src/pwg/mod.rs:72-78) carries the opcode
 * partially evaluated on the instance's witness map (arithmetic.rs:31,38-42; for Opcode::Brillig the offending input expression as written,
 * brillig.rs:46-74), and a binding rebuilds that variant from it instead of from a Default::default(). Returns 1 and fills *head and the
 * arrays when `instance` failed with ACVM_ERR_TOO_MANY_UNKNOWNS, 0 (head zeroed) when it did not. Expression (acir/src/native_types/
 * expression/mod.rs:17-28) = sum mul_coef[i] * w[mul_witnesses[2 i]] * w[mul_witnesses[2 i + 1]] + sum lin_coef[i] * w[lin_witnesses[i]] + q_c,
 * terms in the order the reference's evaluate() leaves them (mul terms that stay, then mul terms that folded into linear ones, then the
 * linear terms); coefficients canonical 32-byte big-endian. Up to cap_mul / cap_lin terms are written (any array may be NULL); head has the counts.
 */
typedef struct {
    uint32_t n_mul, n_lin;
    uint32_t opcode_index; /* the opcode that could not be solved */
    uint8_t q_c[32];
} acvm_expression_t;
int acvm_batch_error_expression(acvm_batch_t *b, const acvm_circuit_t *c, uint32_t instance, acvm_expression_t *head, uint8_t *mul_coef_be32,
                                uint32_t *mul_witnesses /*[cap_mul][2]*/, uint32_t cap_mul, uint8_t *lin_coef_be32, uint32_t *lin_witnesses,
                                uint32_t cap_lin);
/*
 * Per-instance 32-byte digest of the solved witness map for instances [first, first + n), out32 = [n][32] (SURVEY 8d, config 5:
 * callers that keep only the return witnesses use it to compare whole maps without moving them -- the map of the reference is
 * what ACVM::finalize returns, acvm/src/pwg/mod.rs:176-181). Definition: with the two fixed elements of BN254-Fr
 *     g = Blake2s-256("acvm_amd witness map digest: g"),  h = Blake2s-256("acvm_amd witness map digest: h")
 * (the 32 digest bytes read as a big-endian integer, reduced modulo p),
 *     D = sum over the ASSIGNED witnesses w of ( value_w * g^(w+1) + h^(w+1) )  in BN254-Fr,
 *     digest = Blake2s-256( D as 32 big-endian bytes )          (value_w: FieldElement, acir_field/src/generic_ark.rs:156-406).
 * A polynomial fingerprint of the map (two maps differ in D unless g is a root of their difference: probability < 2^-230 for maps
 * that do not depend on g; it is an integrity check, not a commitment against an adversary who knows g). The h-term tells an
 * unassigned witness from one assigned zero. Linear and order-free on purpose: one field product per witness -- a witness the level
 * kernels keep scaled (DESIGN.md "projective witnesses") is folded into its coefficient -- summed in any order, so the library adds
 * a witness as soon as it exists (ACVM_BATCH_FOLD_DIGEST) and may recycle its row (ACVM_BATCH_REUSE_SLOTS). Works for solved,
 * failed and waiting instances alike (the map as it stands). oracle/binding.py witness_map_digest restates it with Python integers.
 */
int acvm_batch_digest(acvm_batch_t *b, uint32_t first, uint32_t n, uint8_t *out32);
/*
 * The same map hashed AS BYTES (ABI 6): SURVEY 8d's "blake2s over the full witness vector", collision-resistant, in tree form so that a map of a
 * million witnesses is not one sequential chain:
 *     leaf_k = Blake2s-256( for w in [256 k, min(256 k + 256, nw)):  value_w as 32 big-endian bytes if the instance assigned w, else 0xFF x 32 )
 *     digest = Blake2s-256( u32_le(nw)  ||  leaf_0  ||  leaf_1  || ... ),        nw = acvm_circuit_num_witnesses
 * (0xFF x 32 is no canonical field element, so an unassigned witness cannot be mistaken for a value.) About as much device work as the solve of a
 * 10^6-opcode tile itself (557 k compressions and 1.1 M canonicalising products per instance) and it needs every row: for audits -- ACVM_E_STATE with
 * ACVM_BATCH_REUSE_SLOTS; acvm_batch_digest above is the one that folds into the solve. oracle/binding.py witness_map_blake2s restates it with hashlib.
 */
int acvm_batch_digest_blake2s(acvm_batch_t *b, uint32_t first, uint32_t n, uint8_t *out32);
/*
 * extract_indices (acvm_js/src/public_witness.rs:10-21; getReturnWitness / getPublicParametersWitness / getPublicWitness
 * with the sets above): values of the listed witnesses for instances [first, first + n), values_be32 = [n][n_witnesses][32].
 * Fails with ACVM_E_STATE and "Failed to extract witness W from witness map. Witness not found." (instance appended) when
 * one of them is unassigned in one of the instances.
 */
int acvm_batch_extract_witnesses(acvm_batch_t *b, const uint32_t *witnesses, uint32_t n_witnesses, uint32_t first, uint32_t n,
                                 uint8_t *values_be32);

/*
 * WitnessMap wire format (acir/src/native_types/witness_map.rs:108-146; acvm_js compressWitness / decompressWitness):
 * gzip(bincode(BTreeMap<Witness, FieldElement>)), a FieldElement being its 64-character hex string.
 * decode: writes up to cap (index, canonical 32-byte big-endian value) pairs in ascending index order, returns the number of
 *         entries of the map (values are reduced like FieldElement::from_hex). Raw bincode without the gzip layer is accepted.
 * encode: returns the number of bytes of the serialised map (written if it fits cap). Same map, same bincode bytes as the
 *         reference; the gzip layer is zlib's at level 9 (any inflater reads it; the compressed bytes are not pinned).
 * acvm_batch_witness_map_bytes: the witness map of one instance as it stands (finalize() for a solved instance, the partial
 *         map otherwise), serialised.
 */
long long acvm_witness_map_decode(const uint8_t *bytes, size_t len, uint32_t *ids, uint8_t *values_be32, uint32_t cap);
long long acvm_witness_map_encode(const uint32_t *ids, const uint8_t *values_be32, uint32_t n, uint8_t *out, size_t cap);
long long acvm_batch_witness_map_bytes(acvm_batch_t *b, uint32_t instance, uint8_t *out, size_t cap);


/*
 * The Rust call shape for ONE instance: struct ACVM (acvm/src/pwg/mod.rs:129-143) as a handle over a batch of one,
 * so that an `impl` of ACVM in Rust forwards method by method.
 *   acvm_new                      ACVM::new(backend, opcodes, initial_witness)  mod.rs:146-156 (opcodes = the circuit's)
 *   acvm_solve / acvm_solve_opcode  ACVM::solve :236-241 / ::solve_opcode :243-303; both return the ACVM_STATUS_* reached
 *   acvm_get_status               the ACVMStatus incl. the OpcodeResolutionError (acvm_result_t)
 *   acvm_instruction_pointer      :171
 *   acvm_witness_map              ACVM::witness_map :161: up to cap (index, 32-byte big-endian value) pairs ascending; returns the map's size
 *   acvm_finalize                 ACVM::finalize :176-181: the same, but ACVM_E_STATE unless the status is Solved (the reference panics)
 *   acvm_get_pending_foreign_call / acvm_resolve_pending_foreign_call   :203-228, same conventions as the batch calls
 */
typedef struct acvm_instance acvm_t;
acvm_t *acvm_new(const acvm_circuit_t *c, const acvm_bb_solver_t *backend, const uint32_t *initial_ids, const uint8_t *values_be32,
                 uint32_t n_initial);
void acvm_free(acvm_t *a);
int acvm_solve(acvm_t *a);
int acvm_solve_opcode(acvm_t *a);
int acvm_get_status(acvm_t *a, acvm_result_t *out);
uint32_t acvm_instruction_pointer(acvm_t *a);
long long acvm_witness_map(acvm_t *a, uint32_t *ids, uint8_t *values_be32, uint32_t cap);
long long acvm_finalize(acvm_t *a, uint32_t *ids, uint8_t *values_be32, uint32_t cap);
int acvm_get_pending_foreign_call(acvm_t *a, acvm_foreign_call_info_t *info);
int acvm_pending_foreign_call_inputs(acvm_t *a, uint32_t *lens, uint8_t *values_be32);
int acvm_resolve_pending_foreign_call(acvm_t *a, uint32_t n_values, const uint8_t *is_array, const uint32_t *lens, const uint8_t *values_be32);

/*
 * ACVM::new takes ANY WitnessMap (mod.rs:146); acvm_batch_new wants one id set for all instances because the circuit is
 * levelised against it. acvm_multi_* lifts that: instance i assigns the ids[offsets[i] .. offsets[i + 1]) (values_be32 in the
 * same order, 32 bytes each); instances with the same id SET share one levelised batch, results come back in the caller's
 * instance order. Everything else is the batch API with an instance index.
 */
typedef struct acvm_multi acvm_multi_t;
acvm_multi_t *acvm_multi_new(const acvm_circuit_t *c, const acvm_bb_solver_t *solver, uint32_t n_instances, const uint64_t *offsets /*[n + 1]*/,
                             const uint32_t *ids, const uint8_t *values_be32);
void acvm_multi_free(acvm_multi_t *m);
uint32_t acvm_multi_num_groups(const acvm_multi_t *m);
int acvm_multi_solve(acvm_multi_t *m); /* number of instances not Solved */
int acvm_multi_results(acvm_multi_t *m, acvm_result_t *out /*[n_instances]*/);
/* witness map of one instance: assigned [n_witnesses], values_be32 [n_witnesses][32]; n_witnesses = acvm_multi_num_witnesses */
uint32_t acvm_multi_num_witnesses(const acvm_multi_t *m);
int acvm_multi_witness_map(acvm_multi_t *m, uint32_t instance, uint8_t *assigned, uint8_t *values_be32);
/* the batch handle and the index inside it that serve `instance` (foreign calls, digests, ... go through the batch API) */
acvm_batch_t *acvm_multi_locate(acvm_multi_t *m, uint32_t instance, uint32_t *index_in_batch);


/*
 * The node-level driver: ONE call solves a global batch of instances of one circuit on every GPU of the node. It stands
 * for the loop a caller of the reference runs once per instance -- ACVM::new, solve, finalize or the error string, the return
 * witnesses (acvm_js/src/execute.rs:60-119, public_witness.rs:10-21) -- and owns what that loop needs on this hardware: the split of the
 * batch over the devices (contiguous, no exchange step), tiles inside a device (a 10k-gate circuit at 2^20 instances is 335 GB of
 * witness table), pinned staging with the upload of tile k + 1 beside the solve of tile k, the exact re-solve of diverging instances
 * beside the next tile, one host thread + one handle + one stream set per device.
 *   acvm_node_new    one batch handle of tile_instances instances per listed device (a device may be listed twice: two handles driven by
 *                    two host threads). keep_ids: the witnesses returned per instance (normally ACVM_SET_RETURN_VALUES).
 *   acvm_node_solve  values_be32 [n_instances][n_initial][32] in host memory (pageable is fine); any of the outputs may be NULL:
 *                    results [n_instances], kept_be32 [n_instances][n_keep][32] (zeros where unassigned), kept_assigned [n_instances][n_keep],
 *                    digests32 [n_instances][32] (acvm_batch_digest's definition). Returns the number of instances not Solved, or a
 *                    negative error. The handle is reusable: call it again with the next global batch.
 */
typedef struct acvm_node acvm_node_t;
typedef struct {
    uint32_t n_devices;       /* 0 = every visible device */
    const int *devices;       /* n_devices indices, NULL = 0 .. n_devices - 1 */
    uint32_t tile_instances;  /* instances per handle; 0 = the largest power of two <= 2^17 whose tables fit the device */
    uint32_t batch_flags;     /* ACVM_BATCH_* of the handles */
} acvm_node_opts_t;
typedef struct {
    uint32_t n_devices, tile_instances;
    uint64_t n_instances;
    double total_ms;            /* wall clock of the last acvm_node_solve */
    int device[16];
    uint32_t async_exact[16];   /* 1: the handle re-solves diverging instances beside the next tile */
    uint32_t tiles[16], exact_instances[16];
    double lane_ms[16];         /* wall clock of the device's thread */
    double solve_device_ms[16]; /* HIP-event time of its solves */
    double h2d_wait_ms[16];     /* what of the uploads the solves did not cover */
    double export_ms[16];       /* results, kept witnesses and digests leaving the device */
    /* host placement (ABI 5): the device's NUMA node (/sys/bus/pci/devices/<bus id>/numa_node, -1 unknown) and the CPUs its lane's two host
     * threads are pinned to (local_cpulist: how many, and the first one; 0 / -1 = not pinned) */
    int numa_node[16];
    uint32_t n_cpus_pinned[16];
    int first_cpu[16];
    /* creation (ABI 6): how often acvm_node_new levelised the circuit (1, or 0 when the circuit's plan cache already held this plan -- the plan
     * is immutable and shared by every lane's handle), its wall clock, the planner's own time, and the process's resident set at the call */
    uint32_t plans_built;
    double create_ms, plan_ms;
    uint64_t host_rss_bytes;
} acvm_node_stats_t;
acvm_node_t *acvm_node_new(const acvm_circuit_t *c, const acvm_bb_solver_t *solver, const uint32_t *initial_ids, uint32_t n_initial,
                           const uint32_t *keep_ids, uint32_t n_keep, const acvm_node_opts_t *opts);
void acvm_node_free(acvm_node_t *n);
uint32_t acvm_node_tile_instances(const acvm_node_t *n);
uint32_t acvm_node_num_devices(const acvm_node_t *n);
long long acvm_node_solve(acvm_node_t *n, uint64_t n_instances, const uint8_t *values_be32, acvm_result_t *results, uint8_t *kept_be32,
                          uint8_t *kept_assigned, uint8_t *digests32);
int acvm_node_stats(acvm_node_t *n, acvm_node_stats_t *out);
/* host-only probes of the lanes' placement logic: a sysfs cpulist ("0-15,32-47") into CPU numbers (returns how many; the first `cap` are
 * written), and the NUMA node + local CPUs of PCI device `bus_id` under `pci_root` (normally /sys/bus/pci/devices) */
int acvm_debug_cpulist(const char *text, uint32_t *cpus, uint32_t cap);
int acvm_debug_device_locality(const char *pci_root, const char *bus_id, int *numa_node, uint32_t *cpus, uint32_t cap);

#ifdef __cplusplus
}
#endif
#endif
