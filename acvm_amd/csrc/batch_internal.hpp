// batch_internal.hpp -- what the translation units of the batch handle share (batch.cpp, batch_schedule.cpp, batch_exact.cpp, batch_export.cpp,
// probes.cpp): staging, the exact path's building blocks, result formatting. Internal to the library; the node driver sees batch.hpp only.
#pragma once
#include "batch.hpp"

template <class T>
static int upload(T **dst, const std::vector<T> &src) {
    size_t bytes = (src.size() ? src.size() : 1) * sizeof(T);
    HIPCHK(hipMalloc((void **)dst, bytes));
    if (!src.empty()) HIPCHK(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}
static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// ---- batch.cpp
// staging arena: `bytes` of device memory valid until the next stage_reserve of this batch (256-byte aligned carving by the caller)
int stage_reserve(acvm_batch *b, size_t bytes);
// forget every resolved foreign-call result (a new ACVM: set_initial_witness / reset)
void clear_fc_store(acvm_batch *b);
void plan_stats(const Plan &p, acvm_stats_t *out);

// ---- batch_schedule.cpp
// ACVM::solve for the batch; next_inputs: acvm_batch_solve_then_import
int batch_solve_impl(acvm_batch *b, const void *next_inputs);

// ---- batch_exact.cpp
int ensure_slow_capacity(acvm_batch *b, uint32_t n);
ExactLanes exact_lanes(acvm_batch *b, uint32_t n_slow);
int upload_fc_tables(acvm_batch *b, uint32_t n_slow);
int run_host_blackbox(acvm_batch *b, uint32_t opcode, bool exact, uint32_t n_slow);
int resolve_internal_calls(acvm_batch *b);  // the caller's BlackBoxFunctionSolver inside Brillig programs: answers the lanes waiting at one; > 0: solve again
// the exact in-order kernels over the current lanes from opcode min_start on (stepping: only opcodes [min_start, end_opcode), nothing replayed)
int run_exact_segments(acvm_batch *b, uint32_t n_slow, uint32_t min_start, bool replay = true, uint32_t end_opcode = 0xFFFFFFFFu);
int retry_device_limits(acvm_batch *b, uint32_t n_slow, bool replay, uint32_t end_opcode);
int count_not_solved(acvm_batch *b);
int solve_resume(acvm_batch *b);
int solve_stepping(acvm_batch *b, bool one);
int ensure_side_table(acvm_batch *b, uint32_t n_lanes, bool own_scratch);
int side_table_outcome(acvm_batch *b, ExactOutcome *out);

// ---- batch_export.cpp
int ensure_digest_tables(acvm_batch *b);
int digest_range(acvm_batch *b, hipStream_t s, const uint4 *W, uint64_t Bp, uint32_t first, uint32_t n, const Unscale &u, const int32_t *d_slow_index,
                 bool use_host_index, uint32_t n_slow, uint8_t *out32);
void format_message(acvm_batch *b, uint32_t j, const SlowResult &sr, acvm_result_t &r);
void fill_result(acvm_batch *b, uint32_t j, acvm_result_t &r);
