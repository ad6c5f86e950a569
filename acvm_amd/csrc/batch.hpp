// batch.hpp -- the batch handle behind the C ABI (include/acvm_amd.h), shared by batch.cpp (the ABI and the solve) and node.cpp (the
// node-level driver: one handle per device). Internal to the library.
#pragma once
#include "../../include/acvm_amd.h"
#include "kernels.hpp"
#include "plan.hpp"
#include "schedule.hpp"
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

using namespace acvm;

// error text of the calling thread (acvm_last_error)
int set_err(int code, const std::string &msg);
#define HIPCHK(expr)                                                                                      \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess)                                                                             \
            return set_err(ACVM_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e));             \
    } while (0)

// Nothing unwinds through the extern "C" boundary: entry points that allocate by input-dependent sizes are function-try-blocks
#define ABI_CATCH                                                                                          \
    catch (const std::bad_alloc &) { return set_err(ACVM_E_NOMEM, "out of host memory"); }                 \
    catch (const std::exception &e) { return set_err(ACVM_E_INVALID, std::string("internal error: ") + e.what()); }
#define ABI_CATCH_PTR                                                                                      \
    catch (const std::bad_alloc &) { set_err(ACVM_E_NOMEM, "out of host memory"); return nullptr; }        \
    catch (const std::exception &e) { set_err(ACVM_E_INVALID, std::string("internal error: ") + e.what()); return nullptr; }

// What the exact path produced for the instances of ONE solve that left the generic path, when it ran beside the NEXT solve's level
// schedule (acvm_batch::async_exact, node.cpp): delivered one solve later (or by batch_finish_pending after the last one).
struct ExactOutcome {
    std::vector<uint32_t> instance;  // index inside that solve's batch
    std::vector<acvm_result_t> results;
    std::vector<uint8_t> kept_values, kept_assigned, digests;  // [n][n_keep][32], [n][n_keep], [n][32]
    void clear() { instance.clear(); results.clear(); kept_values.clear(); kept_assigned.clear(); digests.clear(); }
};

// One plan per (initial witness ids, options, tuning) of a circuit: planning a 10^6-opcode circuit takes seconds and its plan hundreds of MB,
// and the reference's callers build their opcode list once per circuit for any number of executions (acvm_js/src/execute.rs:60-119).
struct PlanKey {
    std::vector<uint32_t> ids, keep;
    bool host_blackbox = false, fold_digest = false, reuse_slots = false;
    std::vector<int64_t> tuning;  // the whole tuning snapshot (every knob may change the plan or the schedule)
    bool operator==(const PlanKey &o) const {
        return ids == o.ids && keep == o.keep && host_blackbox == o.host_blackbox && fold_digest == o.fold_digest && reuse_slots == o.reuse_slots && tuning == o.tuning;
    }
};
struct acvm_circuit {
    std::unique_ptr<Circuit> c;
    // the plans of this circuit (weak: a plan lives as long as a handle uses it; the most recent ones are also held strongly so that
    // create / free / create of handles -- bench legs, the node driver's auto_tile followed by its lanes -- does not plan again)
    mutable std::mutex plan_mutex;
    mutable std::vector<std::pair<PlanKey, std::weak_ptr<const Plan>>> plan_cache;
    mutable std::vector<std::shared_ptr<const Plan>> recent_plans;  // newest last; at most 8, at most ~1 GiB of plan words, always the newest
    mutable uint64_t n_plans_built = 0, n_plans_shared = 0;
};
// the plan of `c` for these options: from the circuit's cache, or built (and cached) now. Never null; a refused circuit's plan has `unsupported` set.
std::shared_ptr<const Plan> plan_for(const acvm_circuit *c, const uint32_t *initial_ids, uint32_t n_initial, const acvm::PlanOpts &opts);

struct acvm_batch {
    // The static plan: immutable once built and shared -- by the handles of a node (node.cpp: one plan per circuit, not one per device) and
    // by every handle created for the same (circuit, initial ids, options, tuning) through the circuit's plan cache (batch.cpp plan_for).
    std::shared_ptr<const Plan> plan_ref;
    const Plan &plan() const { return *plan_ref; }
    LaunchLayout layout;     // where the plan's records are cut into launches for this handle's instance stride (schedule.hpp)
    LevelSchedule schedule;  // the level schedule of one solve: what enqueue_level_schedule walks and what the hazard checker proves
    uint32_t B = 0;          // live instances: what the next import / solve covers (batch_set_live_count; <= capacity)
    uint32_t capacity = 0;   // instances the handle was created for: every table is sized by it
    uint64_t Bp = 0;  // instance stride, multiple of 64
    int device = 0;
    hipStream_t stream = nullptr;
    uint4 *d_W = nullptr, *d_Mem = nullptr;
    uint32_t *d_gate_stream = nullptr, *d_gate_offset = nullptr, *d_consts = nullptr;
    uint32_t *d_prog = nullptr, *d_prog_offset = nullptr, *d_bytecode = nullptr, *d_init_ids = nullptr, *d_producer = nullptr;
    uint32_t *d_dyn_offset = nullptr, *d_slow_start = nullptr;
    uint8_t *d_prog_class = nullptr;  // OpClass per opcode, for the one-launch exact path
    uint32_t *d_cls_offset[N_CLS] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    uint32_t *d_cls_scratch_off[N_CLS] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    uint32_t *d_cls_scratch[N_CLS] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    DeviceProgram dp{};
    uint32_t *d_event = nullptr;       // the event words; EVENT_HDR_WORDS in front of them: the count of flagged instances and the device address of h_flag_count
    uint32_t *d_event_base = nullptr;  // (the allocation)
    uint32_t *h_flag_count = nullptr;  // pinned, device-mapped: the same count where the host reads it after a synchronisation
    bool events_fresh = false;         // the last import left the event words "nobody flagged": the next solve skips its reset launch
    std::vector<uint32_t> h_event;
    bool events_clean = false;  // h_event is all 0xFFFFFFFF, slow_ids empty, slow_index all -1 (kept across solves that flag nothing)
    // exact in-order path
    std::vector<uint32_t> slow_ids, slow_start;
    std::vector<int32_t> slow_index;  // per instance: index into slow_ids or -1
    std::vector<SlowResult> slow_res;
    uint32_t *d_slow_ids = nullptr, *d_assigned = nullptr;
    SlowResult *d_slow_res = nullptr;
    uint32_t slow_cap = 0, n_words = 0;
    bool inputs_set = false, solved = false, force_slow = false, profiling = false;
    hipEvent_t ev_start = nullptr, ev_end = nullptr;
    std::vector<hipEvent_t> ev_pool;
    double solve_device_ms = 0, arith_kernel_ms = 0, dyn_kernel_ms = 0, slow_path_ms = 0;
    double cls_kernel_ms[N_CLS] = {0, 0, 0, 0, 0, 0, 0};
    hipStream_t stream_dyn = nullptr, stream_heavy = nullptr, stream_heavy2 = nullptr, stream_heavy3 = nullptr, stream_digest = nullptr;
    PlanOpts opts;                    // acvm_batch_new_ex: folded digest, slot reuse
    uint4 *d_leaves = nullptr;        // fold_digest: the partial sums of the digest, [records][2][Bp] x 16 B, written by the digest lane during the solve
    uint32_t *d_fp_g = nullptr, *d_fp_gs = nullptr, *d_fp_h = nullptr, *d_fp_hgen = nullptr;  // tables of the digest (kernels.hpp DigestTables), built at the first use
    DigestTables fp{};
    uint32_t *d_slot_of = nullptr;    // reuse_slots: witness -> row of d_W
    // reuse_slots: the exact path re-solves the flagged instances from their initial witnesses in a table of its own (row = witness
    // index, lane t = the t-th flagged instance); x_cap lanes allocated
    uint4 *d_Wx = nullptr, *d_Memx = nullptr;
    uint32_t *d_init_rows = nullptr, *d_ids_x = nullptr;
    uint32_t *d_byte_plane_of = nullptr, *d_byte_plane_of_input = nullptr, *d_byte_plane = nullptr;  // plan.hpp "Byte planes"
    uint64_t x_cap = 0;
    bool reuse() const { return opts.reuse_slots; }
    // Exact path beside the next solve (node.cpp): the flagged instances of a solve are re-solved from their initial witnesses in the
    // side table (d_Wx, like slot reuse) on a stream of their own while the caller loads and solves the next tile; the outcome is
    // collected at the start of that next solve (or by batch_finish_pending).
    bool async_exact = false;          // mode of the handle (batch_enable_async_exact)
    bool pending = false;              // an exact job is in flight on stream_x; slow_ids / slow_index / slow_res describe IT
    bool side_job = false;             // the current exact job works on the side table although the batch does not recycle rows
    hipStream_t stream_x = nullptr;
    hipEvent_t ev_x_ready = nullptr;
    uint32_t *d_x_scratch[N_CLS] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // per-class scratch of the side table's lanes
    uint64_t x_scratch_lanes = 0;
    std::vector<uint32_t> async_keep;  // witnesses delivered with an outcome
    bool async_digest = false;
    ExactOutcome last_outcome;         // of the job a public entry point had to wait for
    bool side() const { return reuse() || side_job; }
    // the table the exact kernels work on
    uint4 *xW() const { return side() ? d_Wx : d_W; }
    uint64_t xBp() const { return side() ? x_cap : Bp; }
    uint32_t *xids() const { return side() ? d_ids_x : d_slow_ids; }
    hipStream_t xstream() const { return pending ? stream_x : stream; }
    uint32_t *xscratch(int cls) const { return pending ? d_x_scratch[cls] : d_cls_scratch[cls]; }
    DeviceProgram xdp() const {
        DeviceProgram d = dp;
        if (side()) { d.Mem = d_Memx; d.slot_of = nullptr; }
        return d;
    }
    std::vector<hipEvent_t> ev_heavy;  // per level 4 events: [4L + q] the records of heavy lane q at the level have run (q < 3)
    std::vector<hipEvent_t> ev_sync;
    uint32_t *d_unscale_index = nullptr, *d_unscale_consts = nullptr, *d_unscale_plain = nullptr, *d_scaled_ids = nullptr;  // projective witnesses (plan.cpp)
    Unscale unscale{};
    uint32_t *d_ped_seed = nullptr;  // seed table of the level Pedersen kernel (one row per Pedersen record)
    uint4 *d_inv = nullptr;  // inverse table: [plan.n_inverse_slots][2 halves][Bp] x 16 B
    uint32_t n_launches = 0;
    // the next tile's import behind this solve (acvm_batch_solve_then_import): the caller's device buffer, and whether the import ran
    const void *next_inputs = nullptr;
    bool next_imported = false;
    hipEvent_t ev_counted = nullptr;  // behind the event count of a solve: what the host waits for instead of the whole stream
    bool holds_tables = false;  // a reference on the device's lookup-table set (grumpkin_host.hpp device_tables_retain)
    // caller-supplied BlackBoxFunctionSolver
    bool has_solver = false;
    acvm_bb_solver_t solver{};
    std::map<uint32_t, std::string> host_bb_msg;  // per instance: error text of a failing callback
    std::map<uint32_t, std::string> fc_fail_msg;  // per instance: the same for a callback made from inside a Brillig program (kept across the re-solves of one input set)
    // Brillig foreign-call round trip (exact lanes only)
    // (fail != 0: not a value but the outcome of a failing INTERNAL call -- the caller's BlackBoxFunctionSolver inside a Brillig program --
    // 1 Failed, 2 Unsupported, 3 panic: the only element of its result; the VM fails there with the text kept in fc_fail_msg)
    struct FcValue { bool is_array; std::vector<FrH> vals; uint32_t fail = 0; };
    struct FcLaneState { bool resolved_new = false; };
    std::vector<FcLaneState> fc_lane;  // per exact lane: the host answered its pending call since the last solve
    // results the host resolved, per Brillig opcode with a ForeignCall (plan.fc_slot_opcode) and per INSTANCE: they accumulate like
    // Brillig::foreign_call_results (pwg/mod.rs:220-224) and serve the level kernels and the exact kernels alike
    struct FcSlot {
        std::map<uint32_t, std::vector<std::vector<FcValue>>> inst;  // instance -> results so far
        uint32_t desc_words = 0, vals_cap = 0;
        uint32_t *d_desc = nullptr;
        uint4 *d_vals = nullptr;
        bool dirty = false;
    };
    std::vector<FcSlot> fc_slots;
    FcStoreSlot *d_fc_store = nullptr;
    uint32_t *d_fc_pend_desc = nullptr;
    uint4 *d_fc_pend_vals = nullptr;
    uint32_t fc_pend_desc_words = 0, fc_pend_vals_cap = 0, fc_lanes_cap = 0;
    std::vector<uint32_t> h_pend_desc, h_pend_vals;
    bool pend_host_valid = false;
    // grow-only device staging arena of the entry points that move data in or out (no hipMalloc / hipFree per call)
    uint8_t *d_stage = nullptr;
    size_t stage_cap = 0;
    uint8_t *d_fetch = nullptr;  // 512 bytes for single-value fetches (message texts): usable while d_stage serves another stream
    // acvm_batch_solve_opcode: every instance is an exact lane, slow_start[t] is its instruction pointer
    bool stepping = false;
    // Brillig retry passes of the exact path (retry_device_limits): the compact VM scratch of the lanes being retried and their columns
    uint32_t *d_br_scratch = nullptr, *d_br_lane = nullptr;
    size_t br_scratch_bytes = 0;
    uint32_t br_lane_cap = 0;
    bool br_retry_active = false;
    uint32_t br_max_regs = 1;  // most registers any Brillig opcode of the circuit uses
    uint32_t n_brillig_retries = 0;  // retry passes of the last solve

    ~acvm_batch() {
        hipSetDevice(device);
        for (void *p : {(void *)d_W, (void *)d_Mem, (void *)d_gate_stream, (void *)d_gate_offset, (void *)d_consts, (void *)d_prog,
                        (void *)d_prog_offset, (void *)d_bytecode, (void *)d_init_ids, (void *)d_producer, (void *)d_dyn_offset,
                        (void *)d_slow_start, (void *)d_event_base, (void *)d_slow_ids, (void *)d_assigned, (void *)d_slow_res})
            if (p) hipFree(p);
        if (h_flag_count) hipHostFree(h_flag_count);
        for (int k = 0; k < (int)N_CLS; k++)
            for (void *p : {(void *)d_cls_offset[k], (void *)d_cls_scratch_off[k], (void *)d_cls_scratch[k]})
                if (p) hipFree(p);
        for (auto e : ev_pool) hipEventDestroy(e);
        for (auto e : ev_sync) hipEventDestroy(e);
        for (auto e : ev_heavy) hipEventDestroy(e);
        if (stream_heavy) hipStreamDestroy(stream_heavy);
        if (stream_heavy2) hipStreamDestroy(stream_heavy2);
        if (stream_heavy3) hipStreamDestroy(stream_heavy3);
        if (stream_digest) hipStreamDestroy(stream_digest);
        if (d_prog_class) hipFree(d_prog_class);
        if (d_leaves) hipFree(d_leaves);
        for (void *p : {(void *)d_fp_g, (void *)d_fp_gs, (void *)d_fp_h, (void *)d_fp_hgen})
            if (p) hipFree(p);
        if (d_slot_of) hipFree(d_slot_of);
        for (void *p : {(void *)d_Wx, (void *)d_Memx, (void *)d_init_rows, (void *)d_ids_x, (void *)d_byte_plane_of, (void *)d_byte_plane_of_input, (void *)d_byte_plane})
            if (p) hipFree(p);
        if (d_inv) hipFree(d_inv);
        for (void *p : {(void *)d_unscale_index, (void *)d_unscale_consts, (void *)d_unscale_plain, (void *)d_scaled_ids})
            if (p) hipFree(p);
        if (d_ped_seed) hipFree(d_ped_seed);
        if (d_stage) hipFree(d_stage);
        if (d_fetch) hipFree(d_fetch);
        if (stream_x) { hipStreamSynchronize(stream_x); hipStreamDestroy(stream_x); }
        if (ev_x_ready) hipEventDestroy(ev_x_ready);
        for (int k = 0; k < (int)N_CLS; k++)
            if (d_x_scratch[k]) hipFree(d_x_scratch[k]);
        if (d_br_scratch) hipFree(d_br_scratch);
        if (d_br_lane) hipFree(d_br_lane);
        for (void *p : {(void *)d_fc_store, (void *)d_fc_pend_desc, (void *)d_fc_pend_vals})
            if (p) hipFree(p);
        for (auto &sl : fc_slots)
            for (void *p : {(void *)sl.d_desc, (void *)sl.d_vals})
                if (p) hipFree(p);
        if (stream_dyn) hipStreamDestroy(stream_dyn);
        if (ev_start) hipEventDestroy(ev_start);
        if (ev_end) hipEventDestroy(ev_end);
        if (ev_counted) hipEventDestroy(ev_counted);
        if (stream) hipStreamDestroy(stream);
        if (holds_tables) device_tables_unref(device);
    }
};

// ---- internal interface of the batch handle for the node-level driver (node.cpp): what the driver needs beyond the public ABI, as calls --
// node.cpp does not reach into the struct. C linkage only because batch.cpp defines them inside its extern "C" block (they are not part
// of the ABI and not declared in the public header).
extern "C" {
// Turns the handle's exact path asynchronous where the circuit allows it (no caller-supplied solver, no foreign calls, a plan the level
// kernels cover entirely); `keep` and `digests` say what an outcome carries. The side table of the first lanes is allocated here, so that
// a device without room for it says so at creation. Returns 1 if enabled, 0 if the handle stays synchronous.
int batch_enable_async_exact(acvm_batch *b, const uint32_t *keep, uint32_t n_keep, bool digests);
// The next import and solve cover instances [0, n) only, 1 <= n <= the instances the handle was created for (the last, partial tile of a
// shard: lanes beyond n are dead instead of solving copies of some instance). A pending exact job of the previous solve is unaffected.
int batch_set_live_count(acvm_batch *b, uint32_t n);
// acvm_batch_set_initial_witness_device without the wait: the import is enqueued on the handle's stream and `imported` recorded behind it;
// the caller keeps d_values_be32 untouched until that event has fired.
int batch_import_async(acvm_batch *b, const void *d_values_be32, hipEvent_t imported);
// waits for the exact job in flight (if any) and moves its outcome into *out (cleared first)
int batch_finish_pending(acvm_batch *b, ExactOutcome *out);
// the outcome of the previous solve's exact job, which the last acvm_batch_solve collected on its way (moved into *out; empty if none)
void batch_take_outcome(acvm_batch *b, ExactOutcome *out);
// the instances of the last solve that left the generic path; pending: their exact job still runs (the outcome arrives one solve later)
const std::vector<uint32_t> *batch_exact_instances(const acvm_batch *b);
bool batch_exact_pending(const acvm_batch *b);
// of the first n instances: those whose (synchronous, final) exact lane did not reach Solved
uint32_t batch_exact_unsolved(const acvm_batch *b, uint32_t n);
// witness w is assigned in the generic instance (the planner's assigned set = what every instance of the level kernels has)
bool batch_generic_assigned(const acvm_batch *b, uint32_t w);
// results / kept witnesses / digests of instances [0, n) of the last solve for the instances the LEVEL kernels solved; instances of the
// exact path are left untouched when an exact job is pending (they arrive with its outcome) and filled in otherwise.
// results [n] or null, kept_values [n][n_keep][32] or null, kept_assigned [n][n_keep] or null, digests [n][32] or null
int batch_export_tile(acvm_batch *b, uint32_t n, const uint32_t *keep, uint32_t n_keep, acvm_result_t *results, uint8_t *kept_values, uint8_t *kept_assigned,
                      uint8_t *digests);
// the kept witnesses of instances [0, n) leave the device without being waited for: the export kernel runs on the handle's stream (behind
// the solve, in front of the next import), the copy into h_out on `copy_stream` behind it; `arrived` is recorded behind the copy.
int batch_enqueue_kept(acvm_batch *b, uint32_t n, const uint32_t *d_keep, uint32_t n_keep, uint8_t *d_out, uint8_t *h_out, hipStream_t copy_stream,
                       hipEvent_t exported, hipEvent_t arrived);
// device bytes a handle of `instances` instances of this plan allocates (tables, class scratch, inverse rows, the first side table)
size_t batch_device_bytes(const acvm::Plan &p, const acvm::PlanOpts &opts, uint64_t instances, bool async_exact);
}  // extern "C"
