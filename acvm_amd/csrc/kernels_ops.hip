// kernels_ops.hip -- kernels of the cheap non-arithmetic opcodes (RANGE, AND/XOR, RecursiveAggregation, Quotient,
// ToLeRadix, MemoryInit, MemoryOp; device routines in ops_light.hpp) and the exact in-order span kernel, which runs
// consecutive opcodes of this class -- Arithmetic included -- with the reference's per-instance semantics
// (acvm/src/pwg/mod.rs:243-303) for the instances that left the generic path.
#include "ops_kernel.hpp"
#include "ops_light.hpp"

namespace acvm {

struct LightOp {
    template <class P>
    static __device__ __forceinline__ OpResult run(const P &p, const uint32_t *__restrict__ rec, const DeviceProgram &dp, uint32_t *, SlowResult *, const ExactLanes *, uint32_t) {
        return dispatch_light(p, rec, dp.consts, dp.Mem);
    }
};

// One lane per flagged instance. Opcodes before the lane's event ran generically on exact data: their witness outputs
// are kept (init_assigned_kernel) and only their memory side effects are re-applied here, because a later opcode of
// the level schedule may already have overwritten the cell.
__global__ void __launch_bounds__(64) exact_span_kernel(uint4 *W, uint64_t Bp, DeviceProgram dp, ExactLanes L, uint32_t op_begin, uint32_t op_end,
                                                        uint32_t replay_memory) {
    const uint32_t t = blockIdx.x * 64 + threadIdx.x;
    if (t >= L.n_slow) return;
    if (L.results[t].status != 1u) return;
    const uint64_t j = L.slow_ids[t];
    const uint32_t start = L.start_opcode[t];
    ExactPolicy p{W, Bp, j, L.assigned, L.n_slow, t};
    FastPolicy replay{W, Bp, j};
    // without memory opcodes nothing before the lane's event has to be replayed
    for (uint32_t oi = replay_memory || start < op_begin ? op_begin : start; oi < op_end; oi++) {
        const uint32_t *__restrict__ rec = dp.prog + dp.prog_offset[oi];
        if (oi < start) {
            if (rec[0] == K_MEM_INIT) op_mem_init(replay, rec, dp.Mem);
            else if (rec[0] == K_MEM_OP) op_mem_op(replay, rec, dp.consts, dp.Mem, true);
            continue;
        }
        const OpResult r = dispatch_light(p, rec, dp.consts, dp.Mem);
        if (r.err) { exact_fail(L, t, oi, r); return; }
    }
}

__global__ void exact_init_kernel(ExactLanes L) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= L.n_slow) return;
    SlowResult r;
    r.status = 1u;  // InProgress
    r.err = r.opcode_index = r.aux0 = r.aux1 = r.msg = r.x0 = r.x1 = r.n_call_stack = 0u;
    for (int i = 0; i < 16; i++) r.call_stack[i] = 0u;
    for (int i = 0; i < 8; i++) r.val[i] = 0u;
    L.results[t] = r;
}
__global__ void exact_finish_kernel(ExactLanes L) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= L.n_slow) return;
    if (L.results[t].status == 1u) {  // pwg/mod.rs:275-276
        L.results[t].status = 0u;
        L.results[t].opcode_index = 0u;  // left over from an earlier RequiresForeignCall stop
        L.results[t].x0 = 0u;
    }
}

void launch_light_level(hipStream_t s, uint4 *W, uint64_t Bp, uint32_t B, const DeviceProgram &dp, const uint32_t *offsets, uint32_t n,
                        uint32_t *event) {
    launch_record_level<LightOp, 256>(s, W, Bp, B, dp, offsets, nullptr, n, event, nullptr);
}
void launch_exact_span(hipStream_t s, uint4 *W, uint64_t Bp, const DeviceProgram &dp, const ExactLanes &L, uint32_t op_begin, uint32_t op_end,
                       bool replay_memory) {
    if (!L.n_slow || op_begin >= op_end) return;
    hipLaunchKernelGGL(exact_span_kernel, dim3((L.n_slow + 63) / 64), dim3(64), 0, s, W, Bp, dp, L, op_begin, op_end, replay_memory ? 1u : 0u);
}
void launch_exact_init(hipStream_t s, const ExactLanes &L) {
    if (!L.n_slow) return;
    hipLaunchKernelGGL(exact_init_kernel, dim3((L.n_slow + 255) / 256), dim3(256), 0, s, L);
}
void launch_exact_finish(hipStream_t s, const ExactLanes &L) {
    if (!L.n_slow) return;
    hipLaunchKernelGGL(exact_finish_kernel, dim3((L.n_slow + 255) / 256), dim3(256), 0, s, L);
}

}  // namespace acvm
