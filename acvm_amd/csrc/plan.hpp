// plan.hpp -- static plan of one circuit against one set of initially-assigned witnesses.
//
// The reference interprets opcodes in program order per instance (acvm/src/pwg/mod.rs:236-303). Which
// witness an opcode assigns depends only on WHICH witnesses are assigned, except for the zero-coefficient
// drop in ArithmeticSolver::evaluate (arithmetic.rs:217-221). The planner replays the assigned-set
// bookkeeping once ("generic" instance: no known multiplicand of an unknown is zero, every opcode succeeds),
// folds every constant divisor (-(sum / coeff), arithmetic.rs:56,86,120) into the gate's coefficients and
// levelises the dependency DAG (memory blocks are chained in program order). Instances that leave the generic
// path -- a failing opcode included -- are detected on the device (event word) and re-solved by the exact
// in-order kernels from the event opcode on.
//
// Two device programs come out of it:
//   * the in-order program `prog`: one record per opcode, original expressions, executed by the exact
//     kernels (ExactPolicy) and, for every non-Arithmetic opcode, also by the level kernels (FastPolicy);
//   * the folded gate stream of the Arithmetic opcodes for arith_level_kernel / arith_dyn_level_kernel.
#pragma once
#include "circuit.hpp"
#include "tuning.hpp"
#include <algorithm>
#include <map>
#include <string>
#include <utility>
#include <vector>

namespace acvm {

// coefficient encoding inside the streams
static constexpr uint32_t COEF_ONE = 0xFFFFFFFFu;
static constexpr uint32_t COEF_MINUS_ONE = 0xFFFFFFFEu;
static constexpr uint32_t COEF_ZERO = 0xFFFFFFFDu;  // only for the constant term / dropped terms

enum GateKind : uint32_t { GATE_ASSERT = 0, GATE_SOLVE = 1, GATE_SOLVE_DYN = 2 };

// Gate record in the u32 stream (arith_level_kernel): gate_record.hpp; evaluation: gate_eval.hpp (flags, bounds: plan.cpp).
// Words:
//  w0 = kind | np_mac << 8 | nl_mac << 16   w1 = opcode index (program order)   w2 = output witness slot (SOLVE*)
//  w3 = constant term (coef encoding)       w4 = slot of 1/denominator in the inverse table (SOLVE_DYN)
//  w5 = counts of the unit-coefficient term lists, then the term lists
// Inversion job (inverse_batch_kernel): [denominator witness, opcode index, inverse slot]
static constexpr uint32_t GATE_HDR_WORDS = 5;

static constexpr uint32_t PLAN_HASH_COOP_FLAG = 0x100u;      // PK_HASH function word: byte message, unpacked through LDS by the level kernel
static constexpr uint32_t PLAN_HASH_CHAIN_FLAG = 0x400u;     // ... and the offset of a chain link follows (the record that hashes this digest runs in the same block)
static constexpr uint32_t PLAN_HASH_RANGE_FLAG = 0x200u;     // ... and (RANGE opcode or NONE, bits) per input follow the outputs: byte RANGE checks fused into the hash
static constexpr uint32_t PLAN_HASH_COOP_MAX_BYTES = 1024;   // 256 message words x 64 instances = 64 KiB of message + 4 KiB of digests + 9 KiB of static tables = 77 KiB of the 160 KiB a gfx950 workgroup may hold (the library runs on gfx950 only; kernels_hash.hip sizes each launch by its own longest record and checks the device's limit)
// record kinds of the in-order program (same numbering as ops_common.hpp RecKind)
enum ProgKind : uint32_t {
    PK_ARITH = 0, PK_RANGE = 1, PK_LOGIC = 2, PK_HASH = 3, PK_PEDERSEN = 4, PK_FIXED_BASE = 5, PK_SCHNORR = 6, PK_ZERO_OUT = 7,
    PK_QUOTIENT = 8, PK_TO_LE_RADIX = 9, PK_MEM_INIT = 10, PK_MEM_OP = 11, PK_BRILLIG = 12, PK_ECDSA = 13, PK_PERM_SORT = 14, PK_DIGEST_LEAF = 15, PK_RANGE_MULTI = 16, PK_BRILLIG_SL = 17
};
// kernel classes of the non-arithmetic records
// CLS_PEDERSEN only exists in the level schedule (its own 4-waves-per-instance-group kernel); the exact path and the
// statistics treat a Pedersen record as CLS_GRUMPKIN
// CLS_HOSTBB: Pedersen / FixedBaseScalarMul / SchnorrVerify when the caller supplied its own BlackBoxFunctionSolver: the
// record is executed by host callbacks between two small kernels (batch.cpp run_host_blackbox)
// CLS_DIGEST: the leaves of the witness-map digest folded into the solve (PlanOpts::fold_digest): records behind the opcodes' in `prog`
enum OpClass : uint32_t { CLS_LIGHT = 0, CLS_HASH = 1, CLS_GRUMPKIN = 2, CLS_BRILLIG = 3, CLS_PEDERSEN = 4, CLS_HOSTBB = 5, CLS_ECDSA = 6, CLS_DIGEST = 7, N_CLS = 8 };

struct Plan {
    Tuning tune;                                // the modes this plan was built with (the batch driver reads its own from here too)
    uint32_t n_witnesses = 0;
    uint32_t n_opcodes = 0;
    std::vector<uint32_t> initial_ids;
    // Byte planes: an initial witness that a byte-message hash of the level schedule reads (PLAN_HASH_COOP_FLAG: every input one byte wide) gets a
    // 4-byte copy per instance beside its row -- bits 0-28 the low 29 bits of the canonical value, bit 31 set when the value is a byte -- written by the
    // import (kernels.hip) and read by the hash kernel in place of the 32-byte row (kernels_hash.hip). byte_plane_of: per witness, its plane or NONE
    // (empty when the circuit has none).
    std::vector<uint32_t> byte_plane_of;
    uint32_t n_byte_planes = 0, n_byte_plane_reads = 0;  // planes; inputs of hash records that are read from one (4 bytes instead of a 32-byte row)
    // ---- arithmetic level program
    std::vector<uint32_t> gate_stream;          // all gate records
    std::vector<uint32_t> gate_offset;          // per scheduled ASSERT/SOLVE gate: offset into gate_stream (level-major)
    std::vector<uint32_t> level_start;          // size n_levels + 1, indexes gate_offset
    std::vector<uint32_t> dyn_offset;           // per inversion job (denominator of a SOLVE_DYN gate), level-major
    std::vector<uint32_t> dyn_level_start;      // size n_levels + 1, indexes dyn_offset
    std::vector<uint32_t> level_needs_inverse;  // size n_levels + 1: the latest inversion level (1-based) whose results a gate of level L (1-based index) reads, 0 = none
    // The heavy record classes run on N_HEAVY_LANES lanes of their own (heavy_lane(): Pedersen | Brillig | hashes, Grumpkin, ECDSA | digest leaves), each
    // in order. Sizes n_levels + 1, values = a 1-based level of that lane, 0 = none:
    std::vector<uint32_t> level_needs_heavy[4];   // [lane]: the latest level of the lane whose outputs the main stream's level L reads
    std::vector<uint32_t> inv_needs_heavy[4];     // same for the inversion batch of level L
    std::vector<uint32_t> lane_needs_lane[4][4];  // [q][q']: the latest level of lane q' whose outputs the records of lane q at level L read
    std::vector<uint32_t> lane_needs_main[4];     // [q]: the latest level of the MAIN stream (gates, light records) whose outputs they read
    std::vector<FrH> constants;                 // Montgomery-form circuit constants
    // ---- projective witnesses (plan.cpp): the level kernels keep witness w as scale_w * value wherever only Arithmetic
    // gates touch it, so that a gate's most expensive coefficient becomes 1. Export and the exact path multiply by 1 / scale.
    std::vector<uint32_t> scaled_ids;           // witnesses stored scaled
    std::vector<FrH> unscale;                   // 1 / scale, same order
    std::vector<uint32_t> unscale_index;        // per witness: index into `unscale`, 0xFFFFFFFF = stored as is
    // relaxed rows (gate_eval.hpp): per witness the bound of its stored representative in units of p / 256 (256 = canonical); how many
    // SOLVE gates store their result as it is / after fr29_weak / canonical; the largest bound any record's result reaches
    std::vector<uint32_t> kbound;
    uint32_t n_gate_out_mode[3] = {0, 0, 0};
    uint32_t max_gate_bound = 0;
    // ---- in-order program: one record per opcode
    std::vector<uint32_t> prog;
    std::vector<uint32_t> prog_offset;          // per opcode
    std::vector<uint8_t> prog_class;            // per opcode: OpClass (Arithmetic = CLS_LIGHT)
    std::vector<uint32_t> prog_scratch;         // per opcode: u32 words of per-instance scratch the record needs
    // ---- non-arithmetic level program: per class, offsets into `prog` (level-major) + per-record scratch words
    std::vector<uint32_t> cls_offset[N_CLS];
    std::vector<uint32_t> cls_scratch[N_CLS];
    std::vector<uint32_t> cls_level_start[N_CLS];  // size n_levels + 1
    uint32_t n_levels = 0;
    uint32_t n_digest_segments = 0;             // fold_digest: records of digest leaves (up to 128 pairs of witnesses each)
    // reuse_slots: row of the witness table per witness (0xFFFFFFFF = never written by the level path), rows in total
    std::vector<uint32_t> slot_of;
    uint32_t n_slots = 0;
    uint32_t mem_cells = 0;                     // cells of the per-instance memory table (all blocks)
    std::vector<uint32_t> bytecode;             // Brillig programs (see plan.cpp emit_brillig)
    // ---- bookkeeping for export / failure masking
    std::vector<uint32_t> producer;             // per witness: opcode index that first assigns it, 0xFFFFFFFF if none,
                                                // 0xFFFFFFFE if initial
    // first opcode that the level kernels cannot execute for the generic instance (static failure); every instance
    // is handed to the exact kernels from there. 0xFFFFFFFF if the whole circuit is covered
    uint32_t truncated_at = 0xFFFFFFFFu;
    // statistics
    uint32_t n_fast_gates = 0, n_dyn_gates = 0, max_level_width = 0, n_other_records = 0;
    uint32_t n_gate_pairs = 0;                  // gates fused behind their producer (plan.cpp "gate pairs")
    uint32_t n_brillig_inlined = 0;             // Brillig opcodes the level schedule runs as straight-line light records (plan.cpp)
    uint32_t n_inverse_slots = 0;               // rows of the inverse table (slots are reused once their gate ran)
    uint64_t algorithmic_bytes = 0, arith_algorithmic_bytes = 0, dyn_algorithmic_bytes = 0;
    uint64_t cls_algorithmic_bytes[N_CLS] = {0, 0, 0, 0, 0, 0, 0};
    double plan_ms = 0;
    std::string unsupported;  // non-empty: circuit holds an opcode no kernel implements
    uint32_t n_hash_chained = 0;  // byte-message hashes that run behind the hash whose digest they consume (hash chains)
    bool needs_grumpkin = false;
    bool needs_ecdsa = false;  // an ECDSA opcode or Brillig black box: the batch carries the generator tables (kernels_ecdsa.hip)
    std::vector<std::pair<uint32_t, uint32_t>> pedersen_seeds;  // per Pedersen record: (number of inputs, domain separator)
    // Brillig foreign calls: function name per (opcode << 32 | bytecode index), buffer sizes of the wait / resolve round trip
    bool has_foreign_calls = false;
    std::map<uint64_t, std::string> fc_function;
    std::vector<uint32_t> fc_slot_opcode;       // per slot of the foreign-call result store: the Brillig opcode (kernels.hpp FcStoreSlot)
    uint32_t fc_max_inputs = 0;
    uint64_t fc_pending_vals = 0;  // field elements one pending call can hand to the host (upper bound)
};

// function names of the INTERNAL foreign calls a caller-supplied BlackBoxFunctionSolver turns the Brillig black-box ops into (plan.cpp):
// the batch driver answers them itself; a circuit's own oracle names cannot start with the control character
#define PLAN_FC_INTERNAL_PREFIX "\x01" "bb:"
#define PLAN_FC_INTERNAL_SCHNORR PLAN_FC_INTERNAL_PREFIX "schnorr_verify"
#define PLAN_FC_INTERNAL_PEDERSEN PLAN_FC_INTERNAL_PREFIX "pedersen"
#define PLAN_FC_INTERNAL_FIXED_BASE PLAN_FC_INTERNAL_PREFIX "fixed_base_scalar_mul"

// largest dense witness table the planner accepts (witness indices 0 .. PLAN_MAX_WITNESSES - 1)
static constexpr uint64_t PLAN_MAX_WITNESSES = 1ull << 27;

static constexpr int N_HEAVY_LANES = 4;
inline int heavy_lane(uint32_t cls) { return cls == CLS_PEDERSEN ? 1 : cls == CLS_BRILLIG ? 2 : cls == CLS_DIGEST ? 3 : 0; }

// what a batch asks of its plan beyond the circuit (acvm_batch_new_ex)
struct PlanOpts {
    bool host_blackbox = false;  // the three BlackBoxFunctionSolver functions are served by caller-supplied host callbacks
    // The per-instance digest of the witness map (acvm_batch_digest) is computed DURING the solve: the leaf of every segment of 256
    // witness indices is a record of its own class, scheduled right behind the last witness of the segment on a lane of its own.
    bool fold_digest = false;
    // Witness-slot liveness reuse: a witness occupies a row of the table from the level that writes it to the
    // level of its last reader; rows are recycled. Only the initial witnesses and `keep` stay until the end.
    bool reuse_slots = false;
    std::vector<uint32_t> keep;
};

Plan build_plan(const Circuit &c, const uint32_t *initial_ids, uint32_t n_initial, const PlanOpts &opts = PlanOpts());

// message words per instance (32-bit) of the byte-message hash record at prog[at], or of the longest member of the chain it heads
inline uint32_t hash_record_lds_words(const std::vector<uint32_t> &pg, size_t at) {
    uint32_t words = 0;
    for (;;) {
        const uint32_t n_in = pg[at + 3];
        words = std::max(words, (n_in + 3u) / 4u);
        if (!(pg[at + 2] & PLAN_HASH_CHAIN_FLAG)) break;
        at = pg[pg[at + 6 + 2 * (size_t)n_in + 64 + ((pg[at + 2] & PLAN_HASH_RANGE_FLAG) ? 2 * (size_t)n_in : 0)]];
    }
    return words;
}
// launch group of a CLS_HASH record inside its level: 0 byte message of <= 256 bytes, 1 longer byte message, 2 the scratch-carrying kernel
inline int hash_launch_group(const std::vector<uint32_t> &pg, size_t at) {
    if (pg[at] != PK_HASH || !(pg[at + 2] & PLAN_HASH_COOP_FLAG)) return 2;  // (the class also holds PK_PERM_SORT records, whose word 2 is an element count)
    return hash_record_lds_words(pg, at) <= 64u ? 0 : 1;
}

}  // namespace acvm
