// kernels_ops.hip -- level kernels of the cheap non-arithmetic opcodes (RANGE, AND/XOR, RecursiveAggregation, Quotient,
// ToLeRadix, MemoryInit, MemoryOp, straight-line Brillig; device routines in ops_light.hpp), the bookkeeping kernels of the
// exact path (its one-launch run kernel lives in kernels_brillig.hip, the only translation unit that sees every opcode class)
// and the kernels around a caller-supplied BlackBoxFunctionSolver.
#include "ops_kernel.hpp"
#include "ops_light.hpp"

namespace acvm {

struct LightOp {
    template <class P>
    static __device__ __forceinline__ OpResult run(const P &p, const uint32_t *__restrict__ rec, const DeviceProgram &dp, uint32_t *, SlowResult *, const ExactLanes *, uint32_t) {
        return dispatch_light(p, rec, dp.consts, dp.Mem);
    }
};
// The straight-line Brillig records of a level (ops_light.hpp op_brillig_sl) have a launch of their own on the main stream: their
// executor carries the integer ALU and the field inversion (111 VGPRs against the 78 of the other light records, which keep their
// occupancy), and its register file takes 16 KiB of LDS per block.
struct LightSlOp {
    template <class P>
    static __device__ __forceinline__ OpResult run(const P &p, const uint32_t *__restrict__ rec, const DeviceProgram &dp, uint32_t *, SlowResult *, const ExactLanes *, uint32_t) {
        __shared__ uint32_t regs[BRILLIG_SL_REGS * 8 * LIGHT_SL_BLOCK];
        return op_brillig_sl(p, rec, dp.consts, regs);
    }
};

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
// min_ip: only lanes whose instruction pointer (start_opcode) reached it are done (acvm_batch_solve_opcode; 0 otherwise)
__global__ void exact_finish_kernel(ExactLanes L, uint32_t min_ip) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= L.n_slow) return;
    if (L.results[t].status == 1u && L.start_opcode[t] >= min_ip) {  // pwg/mod.rs:275-276
        L.results[t].status = 0u;
        L.results[t].opcode_index = 0u;  // left over from an earlier RequiresForeignCall stop
        L.results[t].x0 = 0u;
    }
}

// ---------------------------------------------------------------------------------------------- caller-supplied BlackBoxFunctionSolver
// A Pedersen / FixedBaseScalarMul / SchnorrVerify record served by host callbacks (acvm_bb_solver_t): the inputs leave the
// device through hostbb_gather_kernel, the callbacks run on the host, hostbb_apply_kernel inserts their outputs with the
// reference's insert_value semantics. On the exact path hostbb_precheck_kernel first applies the all-inputs-assigned rule
// (blackbox/mod.rs:55-62) and marks the lanes that really execute the opcode.
__global__ void hostbb_precheck_kernel(ExactLanes L, uint32_t opcode, const uint32_t *__restrict__ sel, uint32_t n_sel, uint8_t *__restrict__ active) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= L.n_slow) return;
    active[t] = 0;
    if (L.results[t].status != 1u || L.start_opcode[t] > opcode) return;
    for (uint32_t i = 0; i < n_sel; i++) {
        const uint32_t w = sel[i];
        if (!((L.assigned[(uint64_t)(w >> 5) * L.n_slow + t] >> (w & 31)) & 1u)) {
            exact_fail(L, t, opcode, op_fail(DE_MISSING_ASSIGNMENT, w));
            return;
        }
    }
    active[t] = 1;
}
// out: [n_lanes][n_sel][32] canonical big-endian; lane t reads instance ids[t] (or first + t when ids is null)
// (slot_of: the table's rows under witness-slot reuse, null = the witness index)
__global__ void __launch_bounds__(256) hostbb_gather_kernel(const uint4 *__restrict__ W, uint64_t Bp, const uint32_t *__restrict__ ids, uint32_t first,
                                                            uint32_t n_lanes, const uint32_t *__restrict__ sel, uint32_t n_sel, uint8_t *__restrict__ out,
                                                            const uint32_t *__restrict__ slot_of) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t k = blockIdx.y;
    if (t >= n_lanes) return;
    const uint64_t j = ids ? ids[first + t] : first + t;
    const Fr x = fr_to_canonical(fr_load(W, slot_of ? slot_of[sel[k]] : sel[k], Bp, j));
    uint8_t *p = out + ((uint64_t)t * n_sel + k) * 32;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint8_t *q = p + 28 - 4 * i;
        q[0] = (uint8_t)(x.v[i] >> 24); q[1] = (uint8_t)(x.v[i] >> 16); q[2] = (uint8_t)(x.v[i] >> 8); q[3] = (uint8_t)x.v[i];
    }
}
__device__ __forceinline__ Fr fr_from_be32_reduce(const uint8_t *p) {
    Fr x;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint8_t *q = p + 28 - 4 * i;
        x.v[i] = (uint32_t)q[0] << 24 | (uint32_t)q[1] << 16 | (uint32_t)q[2] << 8 | (uint32_t)q[3];
    }
    return fr_from_canonical(canon_reduce(x));
}
// rc per lane: 0 Ok, 1 BlackBoxResolutionError::Failed, 2 Unsupported (blackbox_solver/src/lib.rs:15-21); 255 = lane not called
template <class P>
__device__ __forceinline__ OpResult hostbb_apply(const P &p, uint32_t func, uint32_t rc, const uint32_t *__restrict__ outs, uint32_t n_out,
                                                 const uint8_t *__restrict__ vals) {
    if (rc == 1u) return op_fail_msg(DE_BLACKBOX_FAILED, func, DM_HOST_MESSAGE);
    if (rc == 2u) return op_fail(DE_UNSUPPORTED_BLACKBOX, func);
    if (rc != 0u) return op_fail_msg(DE_PANIC, func, DM_HOST_MESSAGE);
    for (uint32_t k = 0; k < n_out; k++)
        if (!p.insert(outs[2 * k], fr_from_be32_reduce(vals + 32 * k), outs[2 * k + 1])) return op_fail(DE_UNSATISFIED);
    return op_ok();
}
__global__ void __launch_bounds__(256) hostbb_apply_level_kernel(uint4 *W, uint64_t Bp, uint32_t first, uint32_t n_lanes, uint32_t opcode, uint32_t func,
                                                                 const uint32_t *__restrict__ outs, uint32_t n_out, const uint8_t *__restrict__ rc,
                                                                 const uint8_t *__restrict__ vals, uint32_t *__restrict__ event, const uint32_t *__restrict__ slot_of) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_lanes) return;
    const uint64_t j = first + t;
    FastPolicy p{W, Bp, j, slot_of};
    const OpResult r = hostbb_apply(p, func, rc[t], outs, n_out, vals + (uint64_t)t * n_out * 32);
    if (r.err) flag_instance(event, j, opcode);
}
__global__ void __launch_bounds__(64) hostbb_apply_exact_kernel(uint4 *W, uint64_t Bp, ExactLanes L, uint32_t first, uint32_t n_lanes, uint32_t opcode,
                                                                uint32_t func, const uint32_t *__restrict__ outs, uint32_t n_out,
                                                                const uint8_t *__restrict__ active, const uint8_t *__restrict__ rc,
                                                                const uint8_t *__restrict__ vals) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_lanes) return;
    const uint32_t t = first + i;
    if (!active[t]) return;
    ExactPolicy p{W, Bp, L.slow_ids[t], L.assigned, L.n_slow, t};
    const OpResult r = hostbb_apply(p, func, rc[i], outs, n_out, vals + (uint64_t)i * n_out * 32);
    if (r.err) exact_fail(L, t, opcode, r);
}
void launch_hostbb_precheck(hipStream_t s, const ExactLanes &L, uint32_t opcode, const uint32_t *sel, uint32_t n_sel, uint8_t *active) {
    if (!L.n_slow) return;
    hipLaunchKernelGGL(hostbb_precheck_kernel, dim3((L.n_slow + 255) / 256), dim3(256), 0, s, L, opcode, sel, n_sel, active);
}
void launch_hostbb_gather(hipStream_t s, const uint4 *W, uint64_t Bp, const uint32_t *ids, uint32_t first, uint32_t n_lanes, const uint32_t *sel,
                          uint32_t n_sel, uint8_t *out, const uint32_t *slot_of) {
    if (!n_lanes || !n_sel) return;
    hipLaunchKernelGGL(hostbb_gather_kernel, dim3((n_lanes + 255) / 256, n_sel), dim3(256), 0, s, W, Bp, ids, first, n_lanes, sel, n_sel, out, slot_of);
}
void launch_hostbb_apply_level(hipStream_t s, uint4 *W, uint64_t Bp, uint32_t first, uint32_t n_lanes, uint32_t opcode, uint32_t func, const uint32_t *outs,
                               uint32_t n_out, const uint8_t *rc, const uint8_t *vals, uint32_t *event, const uint32_t *slot_of) {
    if (!n_lanes) return;
    hipLaunchKernelGGL(hostbb_apply_level_kernel, dim3((n_lanes + 255) / 256), dim3(256), 0, s, W, Bp, first, n_lanes, opcode, func, outs, n_out, rc, vals, event, slot_of);
}
void launch_hostbb_apply_exact(hipStream_t s, uint4 *W, uint64_t Bp, const ExactLanes &L, uint32_t first, uint32_t n_lanes, uint32_t opcode, uint32_t func,
                               const uint32_t *outs, uint32_t n_out, const uint8_t *active, const uint8_t *rc, const uint8_t *vals) {
    if (!n_lanes) return;
    hipLaunchKernelGGL(hostbb_apply_exact_kernel, dim3((n_lanes + 63) / 64), dim3(64), 0, s, W, Bp, L, first, n_lanes, opcode, func, outs, n_out, active, rc, vals);
}

void launch_light_level(hipStream_t s, uint4 *W, uint64_t Bp, uint32_t B, const DeviceProgram &dp, const uint32_t *offsets, uint32_t n,
                        uint32_t *event) {
    launch_record_level<LightOp, 256>(s, W, Bp, B, dp, offsets, nullptr, n, event, nullptr);
}
// A level's gates and its light records in ONE launch (batch.cpp): both are independent work of the same level on the main stream, and as
// launches of their own the few light records of a level (540 launches of ~20 us on the 10^6-opcode circuit) each ran alone between the drain
// of one gate launch and the ramp of the next. The light blocks come first (blockIdx.y < n_light): they are the longer ones. Both bodies
// fit the gate kernel's register budget (78 VGPRs each).
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8)))
arith_light_level_kernel(uint4 *__restrict__ W, uint64_t Bp, uint32_t B, const uint32_t *__restrict__ gate_stream, const uint32_t *__restrict__ gate_offset,
                         const uint32_t *__restrict__ consts, uint32_t *__restrict__ event, const uint4 *__restrict__ Inv, uint32_t n_light,
                         const uint32_t *__restrict__ light_offsets, const uint32_t *__restrict__ prog, const uint32_t *__restrict__ slot_of, uint4 *Mem) {
    if (blockIdx.y >= n_light) {
        arith_level_body(W, Bp, B, gate_stream, gate_offset, consts, event, Inv, blockIdx.y - n_light);
        return;
    }
    const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= B) return;
    const uint32_t *__restrict__ rec = prog + light_offsets[blockIdx.y];
    FastPolicy p{W, Bp, j, slot_of};
    const OpResult r = dispatch_light(p, rec, consts, Mem);
    if (r.err) flag_instance(event, j, rec[0] == K_RANGE_MULTI ? r.aux0 : rec[1]);
}
void launch_arith_light_level(hipStream_t s, uint4 *W, uint64_t Bp, uint32_t B, const uint32_t *gate_stream, const uint32_t *gate_offset, uint32_t n_gates,
                              const uint4 *inv, const DeviceProgram &dp, const uint32_t *light_offsets, uint32_t n_light, uint32_t *event) {
    if (!B || !(n_gates + n_light)) return;
    hipLaunchKernelGGL(arith_light_level_kernel, dim3((B + 255) / 256, n_gates + n_light), dim3(256), 0, s, W, Bp, B, gate_stream, gate_offset, dp.consts, event, inv,
                       n_light, light_offsets, dp.prog, dp.slot_of, dp.Mem);
}
void launch_light_sl_level(hipStream_t s, uint4 *W, uint64_t Bp, uint32_t B, const DeviceProgram &dp, const uint32_t *offsets, uint32_t n,
                           uint32_t *event) {
    launch_record_level<LightSlOp, LIGHT_SL_BLOCK>(s, W, Bp, B, dp, offsets, nullptr, n, event, nullptr);
}
void launch_exact_init(hipStream_t s, const ExactLanes &L) {
    if (!L.n_slow) return;
    hipLaunchKernelGGL(exact_init_kernel, dim3((L.n_slow + 255) / 256), dim3(256), 0, s, L);
}
void launch_exact_finish(hipStream_t s, const ExactLanes &L, uint32_t min_ip) {
    if (!L.n_slow) return;
    hipLaunchKernelGGL(exact_finish_kernel, dim3((L.n_slow + 255) / 256), dim3(256), 0, s, L, min_ip);
}

}  // namespace acvm
