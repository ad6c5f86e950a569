// kernels.hpp -- host-visible launchers of the gfx950 kernels (kernels*.hip).
#pragma once
#include "grumpkin_host.hpp"
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace acvm {

// per-instance outcome of the exact in-order kernels. status 1 (InProgress) while the instance is still running.
struct SlowResult {
    uint32_t status, err, opcode_index, aux0, aux1;
    uint32_t msg, x0, x1;  // DevMsg code + payload for the host-side message text
    uint32_t n_call_stack;
    uint32_t call_stack[16];
    uint32_t val[8];       // a canonical field value quoted by the message (Brillig black-box limb checks)
};

// Brillig foreign-call round trip (ACVM::get_pending_foreign_call / resolve_pending_foreign_call, pwg/mod.rs:203-228).
// Results the host resolved live in a batch-wide store, one slot per Brillig opcode that holds a ForeignCall, keyed by the
// INSTANCE (so the level kernels and the exact kernels read the same tables): descriptor words
// [n_results, (n_values, (is_array, n) x n_values) x n_results] of instance j at desc[w * Bp + j], values laid out like W.
struct FcStoreSlot {
    const uint32_t *desc;  // null: nothing resolved for this opcode yet
    const uint4 *vals;
};
// Inputs of a pending call of the exact lanes: pend_desc [n_inputs, len x n_inputs], values in pend_vals, laid out like W with
// stride n_slow.
struct FcLanes {
    uint32_t *pend_desc;
    uint32_t pend_desc_words;
    uint4 *pend_vals;
    uint32_t pend_vals_cap;
};

// limits of the Brillig VM for one launch (ops_brillig.hpp; batch.cpp retry_device_limits raises them on the exact path)
struct BrilligLimits {
    uint32_t steps;       // instructions one VM run may execute
    uint32_t call_depth;  // call-stack words
    uint32_t mem_cap;     // memory cells per lane; 0 = the record's own (planner estimate)
    uint64_t stride;      // retry passes: lanes of the compact scratch (ExactLanes::br_lane indexes them)
};

// lanes of the exact path: flagged instances gathered through slow_ids
struct ExactLanes {
    const uint32_t *slow_ids;
    uint32_t n_slow;
    uint32_t *assigned;              // bit w of lane t at assigned[(w >> 5) * n_slow + t]
    const uint32_t *start_opcode;    // first opcode the lane executes (its event)
    SlowResult *results;
    FcLanes fc;
    const uint32_t *br_lane = nullptr;  // Brillig retry pass: column of lane t in the compact VM scratch (0xFFFFFFFF: not retried); null otherwise
};

// everything a record needs besides the witness table
struct DeviceProgram {
    const uint32_t *prog;         // in-order program (one record per opcode)
    const uint32_t *prog_offset;  // per opcode
    const uint32_t *consts;       // circuit constants, 8 x u32 Montgomery each
    const uint32_t *bytecode;     // Brillig programs
    uint4 *Mem;                   // per-instance memory blocks, laid out like W
    GrumpkinTables grumpkin;      // device lookup tables (null pointers if the circuit has no Grumpkin opcode)
    const uint32_t *ecdsa_g;      // generator tables of secp256k1 / secp256r1 (kernels_ecdsa.hip; null: the circuit has no ECDSA call)
    const uint32_t *ped_seed;     // per Pedersen record: hash_single(x of hash_pair(IV[domain separator], n), 0), affine, 16 x u32
    const FcStoreSlot *fc_store;  // device array, one entry per Brillig opcode with a ForeignCall (null: the circuit has none)
    const uint32_t *slot_of;      // witness -> row of the table for the level kernels (null: row = witness index; plan.cpp slot reuse)
    BrilligLimits brillig;        // limits of the Brillig VM for this launch
    const uint32_t *byte_plane_of; // witness -> its byte plane or NONE (null: the circuit has none; plan.hpp "Byte planes")
    const uint32_t *byte_plane;    // [plane][instance]: low 29 bits of the canonical value | is-byte << 31, written by the import
};

// projective witnesses (plan.cpp): device tables behind the export and the hand-over to the exact path
struct Unscale {
    const uint32_t *index;       // per witness: row of consts, 0xFFFFFFFF = stored as is (null: no witness is scaled)
    const uint32_t *consts;      // 1 / scale, 8 x u32 each (device Montgomery form)
    const uint32_t *consts_plain; // 1 / scale as a canonical integer: the Montgomery product with it is the canonical VALUE (unscale and leave Montgomery form in one)
    const uint32_t *scaled_ids;  // the scaled witnesses, in row order
    uint32_t n_scaled;
    const uint32_t *event;       // per instance: 0xFFFFFFFF = solved by the level kernels (its column is still scaled)
};

// plane_of_input / plane: per input its byte plane or NONE, and the planes (both null: the circuit has none)
// event_reset: null, or the batch's event words: the import also leaves them "nobody flagged" (returns true when it did: the solve behind it needs no reset launch)
bool launch_import(hipStream_t s, uint4 *W, uint64_t Bp, uint32_t B, const uint8_t *in, const uint32_t *ids, uint32_t n_in, const uint32_t *gate = nullptr,
                   const uint32_t *plane_of_input = nullptr, uint32_t *plane = nullptr, uint32_t *event_reset = nullptr);
void launch_export(hipStream_t s, const uint4 *W, uint64_t Bp, uint32_t first, uint32_t n, const uint32_t *sel, uint32_t n_sel,
                   uint8_t *out, const Unscale &u, const uint32_t *row_of = nullptr);
void launch_gather_initial(hipStream_t s, uint4 *Wx, uint64_t Bpx, const uint4 *W, uint64_t Bp, const uint32_t *init_ids, const uint32_t *init_rows, uint32_t n_init,
                           const uint32_t *slow_ids, uint32_t n_slow);
void launch_gather_columns(hipStream_t s, uint4 *Wx, uint64_t Bpx, const uint4 *W, uint64_t Bp, uint32_t n_rows, const uint32_t *slow_ids, uint32_t n_slow,
                           const uint32_t *unscale_index, const uint32_t *unscale_consts, const uint32_t *producer, const uint32_t *start_opcode);
void launch_unscale_slow(hipStream_t s, uint4 *W, uint64_t Bp, const uint32_t *slow_ids, uint32_t n_slow, const Unscale &u, const uint32_t *producer,
                         const uint32_t *start_opcode);
void launch_arith_level(hipStream_t s, uint4 *W, uint64_t Bp, uint32_t B, const uint32_t *gate_stream, const uint32_t *gate_offset,
                        uint32_t n_gates, const uint32_t *consts, uint32_t *event, const uint4 *inv);
void launch_inverse_batch(hipStream_t s, const uint4 *W, uint4 *inv, uint64_t Bp, uint32_t B, const uint32_t *gate_stream,
                          const uint32_t *job_offset, uint32_t n_jobs, uint32_t *event, uint32_t inv_chunk);
// gates [0, n_gates) of a level and n_light light records of the same level in one launch (n_gates + n_light <= 65535)
void launch_arith_light_level(hipStream_t s, uint4 *W, uint64_t Bp, uint32_t B, const uint32_t *gate_stream, const uint32_t *gate_offset, uint32_t n_gates,
                              const uint4 *inv, const DeviceProgram &dp, const uint32_t *light_offsets, uint32_t n_light, uint32_t *event);
void launch_modmul_rate(hipStream_t s, uint32_t *out, uint32_t blocks, uint32_t iters);
void launch_secp_rate(hipStream_t s, uint32_t curve, uint32_t *out, uint32_t blocks, uint32_t iters);  // kernels_ecdsa.hip
void launch_stream_rate(hipStream_t s, const uint4 *a, const uint4 *b, uint4 *out, uint64_t n);  // n: multiple of 256
void launch_fr_selftest(hipStream_t s, uint64_t seed, uint32_t n, uint32_t *mismatches);
void launch_fill_u32(hipStream_t s, uint32_t *p, uint32_t v, uint64_t n);
// event words [0, B) <- 0xFFFFFFFF, the count of flagged instances in front of them (event[-4], ops_common.hpp flag_instance) <- 0
void launch_event_reset(hipStream_t s, uint32_t *event, uint32_t B);
// event words <- min(event, opcode), the count <- B: a plan the level kernels do not cover entirely
void launch_event_truncate(hipStream_t s, uint32_t *event, uint32_t B, uint32_t opcode);
void launch_min_u32(hipStream_t s, uint32_t *p, uint32_t v, uint64_t n);
void launch_init_assigned(hipStream_t s, uint32_t *assigned, uint32_t n_slow, uint32_t n_words, uint32_t n_witnesses,
                          const uint32_t *producer, const uint32_t *start_opcode);

// ---- level kernels of the non-arithmetic record classes (FastPolicy): grid.y = records of the level.
// offsets: device array of record offsets into prog; scratch_off: per record, u32-word offset (per instance) into scratch
void launch_light_level(hipStream_t s, uint4 *W, uint64_t Bp, uint32_t B, const DeviceProgram &dp, const uint32_t *offsets, uint32_t n,
                        uint32_t *event);
// the straight-line Brillig records of a level (PK_BRILLIG_SL, ops_light.hpp op_brillig_sl)
void launch_light_sl_level(hipStream_t s, uint4 *W, uint64_t Bp, uint32_t B, const DeviceProgram &dp, const uint32_t *offsets, uint32_t n,
                           uint32_t *event);
void launch_hash_level(hipStream_t s, uint4 *W, uint64_t Bp, uint32_t B, const DeviceProgram &dp, const uint32_t *offsets,
                       const uint32_t *scratch_off, uint32_t n, uint32_t *event, uint32_t *scratch);
// records flagged HASH_COOP_FLAG (ops_hash.hpp): four waves per 64 instances, the byte message in LDS
// lds_words: 32-bit message words per instance of the longest record of the launch
void launch_hash_coop_level(hipStream_t s, uint4 *W, uint64_t Bp, uint32_t B, const DeviceProgram &dp, const uint32_t *offsets, uint32_t n, uint32_t *event, uint32_t lds_words);
void launch_grumpkin_level(hipStream_t s, uint4 *W, uint64_t Bp, uint32_t B, const DeviceProgram &dp, const uint32_t *offsets,
                           const uint32_t *scratch_off, uint32_t n, uint32_t *event, uint32_t *scratch);
// Pedersen records: 4 waves per group of 64 instances, one accumulator chain each (kernels_grumpkin.hip)
void launch_pedersen_seeds(hipStream_t s, const GrumpkinTables &T, const uint32_t *keys, uint32_t n, uint32_t *out);
void launch_pedersen_level(hipStream_t s, uint4 *W, uint64_t Bp, uint32_t B, const DeviceProgram &dp, const uint32_t *offsets, uint32_t n,
                           uint32_t *event, const uint32_t *scratch_off = nullptr, uint32_t *scratch = nullptr);
void launch_brillig_level(hipStream_t s, uint4 *W, uint64_t Bp, uint32_t B, const DeviceProgram &dp, const uint32_t *offsets,
                          const uint32_t *scratch_off, uint32_t n, uint32_t *event, uint32_t *scratch);

// ---- exact in-order kernels (ExactPolicy): one lane per flagged instance
void launch_exact_init(hipStream_t s, const ExactLanes &L);
// opcodes [op_begin, op_end) of EVERY class but CLS_HOSTBB in one launch (kernels_brillig.hip exact_run_kernel); prog_class: device
// array, OpClass per opcode; the class scratch buffers as the per-opcode kernels take them
struct ExactScratch { uint32_t *hash, *grumpkin, *brillig; };
void launch_exact_run(hipStream_t s, uint4 *W, uint64_t Bp, const DeviceProgram &dp, const ExactLanes &L, uint32_t op_begin, uint32_t op_end, bool replay_memory,
                      const uint8_t *prog_class, const ExactScratch &sc);
void launch_grumpkin_probe(hipStream_t s, const GrumpkinTables &T, uint32_t what, uint32_t param, const uint32_t *in, uint32_t n_in, uint32_t *out);
// ECDSA (kernels_ecdsa.hip; the generator tables d * 2^(16 j) * G of both curves: grumpkin_host.hpp ecdsa_generator_tables)
void launch_secp_probe(hipStream_t s, uint32_t curve, uint32_t what, const uint32_t *in, uint32_t n_items, uint32_t words_in, uint32_t words_out, uint32_t *out);
void launch_ecdsa_level(hipStream_t s, uint4 *W, uint64_t Bp, uint32_t B, const DeviceProgram &dp, const uint32_t *offsets, uint32_t n, uint32_t *event);
// caller-supplied BlackBoxFunctionSolver (kernels_ops.hip)
void launch_hostbb_precheck(hipStream_t s, const ExactLanes &L, uint32_t opcode, const uint32_t *sel, uint32_t n_sel, uint8_t *active);
void launch_hostbb_gather(hipStream_t s, const uint4 *W, uint64_t Bp, const uint32_t *ids, uint32_t first, uint32_t n_lanes, const uint32_t *sel,
                          uint32_t n_sel, uint8_t *out, const uint32_t *slot_of = nullptr);
void launch_hostbb_apply_level(hipStream_t s, uint4 *W, uint64_t Bp, uint32_t first, uint32_t n_lanes, uint32_t opcode, uint32_t func, const uint32_t *outs,
                               uint32_t n_out, const uint8_t *rc, const uint8_t *vals, uint32_t *event, const uint32_t *slot_of = nullptr);
void launch_hostbb_apply_exact(hipStream_t s, uint4 *W, uint64_t Bp, const ExactLanes &L, uint32_t first, uint32_t n_lanes, uint32_t opcode, uint32_t func,
                               const uint32_t *outs, uint32_t n_out, const uint8_t *active, const uint8_t *rc, const uint8_t *vals);
// per-instance digest of the witness map (kernels_hash.hip; definition: include/acvm_amd.h acvm_batch_digest). Device tables of the
// polynomial fingerprint, 8 x u32 per entry in the device's Montgomery form:
struct DigestTables {
    const uint32_t *g_pow;     // [n_witnesses]: g^(w+1)
    const uint32_t *g_scaled;  // [scaled witnesses, rows of Unscale]: g^(w+1) / scale_w
    const uint32_t *h_pow;     // [n_witnesses]: h^(w+1)
    const uint32_t *h_generic; // [1]: the sum of h^(w+1) over the witnesses the planner saw assigned
};
// the witness map hashed as bytes, in tree form (kernels_hash.hip; include/acvm_amd.h acvm_batch_digest_blake2s): leaves = scratch of
// digest_b2s_leaves(n_witnesses) x 8 x n words
uint32_t digest_b2s_leaves(uint32_t n_witnesses);
void launch_digest_blake2s(hipStream_t s, const uint4 *W, uint64_t Bp, uint32_t first, uint32_t n, uint32_t n_witnesses, const uint32_t *producer, const Unscale &u,
                           const int32_t *slow_index, const uint32_t *assigned, uint32_t n_slow, uint32_t *leaves, uint8_t *out);
uint32_t digest_chunks(uint32_t n_witnesses);  // rows of the partial-sum scratch of launch_digest: digest_chunks x n x 32 bytes
void launch_digest(hipStream_t s, const uint4 *W, uint64_t Bp, uint32_t first, uint32_t n, uint32_t n_witnesses, const uint32_t *producer, const Unscale &u,
                   const DigestTables &T, const int32_t *slow_index, const uint32_t *assigned, uint32_t n_slow, uint4 *partial, uint8_t *out);
// the digest folded into the solve (PlanOpts::fold_digest): partial = [records][2][Bp] x 16 B
void launch_digest_fold_level(hipStream_t s, const uint4 *W, uint64_t Bp, uint32_t B, const DeviceProgram &dp, const uint32_t *offsets, uint32_t n,
                              const DigestTables &T, uint4 *partial);
void launch_digest_final(hipStream_t s, const uint4 *partial, uint32_t n_rows, uint64_t stride, uint32_t first, uint32_t n, const uint32_t *event, const DigestTables &T,
                         uint8_t *out);
// InProgress -> Solved after the last opcode
void launch_exact_finish(hipStream_t s, const ExactLanes &L, uint32_t min_ip = 0);

}  // namespace acvm
