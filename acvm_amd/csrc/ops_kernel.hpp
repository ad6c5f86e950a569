// ops_kernel.hpp -- the kernel skeleton every non-arithmetic record class is instantiated in:
//   record_level_kernel<Op>  FastPolicy: grid = (instances / block, records of one dependency level); errors only flag
//                            the instance (event word = min failing opcode index) for the exact path
// Op::run(policy, record, program, per-record scratch) is the templated device routine of the class. The exact path
// (ExactPolicy: per-instance assigned set, errors become the instance's final result) runs every class in ONE kernel,
// kernels_brillig.hip exact_run_kernel.
#pragma once
#include "kernels.hpp"
#include "ops_common.hpp"

namespace acvm {

__device__ __forceinline__ void exact_fail(const ExactLanes &L, uint32_t t, uint32_t opcode, const OpResult &r) {
    SlowResult &o = L.results[t];
    o.status = 2u;  // ACVM_STATUS_FAILURE
    o.err = r.err;
    o.opcode_index = opcode;
    o.aux0 = r.aux0;
    o.aux1 = r.aux1;
    o.msg = r.msg;
    o.x0 = r.x0;
    o.x1 = r.x1;
}

// prog / consts / slot_of also arrive as kernel arguments of their own: only a noalias ARGUMENT lets the compiler read the record, the constants
// and the row map with scalar loads; through the by-value struct every record word is a vector load that waits behind the stores before it.
template <class Op, int BLOCK>
__global__ void __launch_bounds__(BLOCK) record_level_kernel(uint4 *W, uint64_t Bp, uint32_t B, DeviceProgram dp, const uint32_t *__restrict__ offsets,
                                                             const uint32_t *__restrict__ scratch_off, uint32_t *__restrict__ event,
                                                             uint32_t *scratch, const uint32_t *__restrict__ prog, const uint32_t *__restrict__ consts,
                                                             const uint32_t *__restrict__ slot_of, const uint32_t *__restrict__ bytecode) {
    const uint64_t j = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (j >= B) return;
    dp.prog = prog;
    dp.consts = consts;
    dp.slot_of = slot_of;
    dp.bytecode = bytecode;
    const uint32_t *__restrict__ rec = dp.prog + offsets[blockIdx.y];
    // The record's per-lane scratch (scratch_off: offset and words per record) is laid out per WAVE inside the words x Bp region the planner reserved:
    // [wave][word][64 lanes]. A wave's words are consecutive 256-byte lines; laid out [word][instance] they were a power of two (4 x instances
    // bytes) apart, and the 432-word window table of SchnorrVerify kept landing on the same few channels: config 4 ran at 2.0 or at 3.3 ms per 2^16
    // depending on where the process's buffers happened to sit (profiles/r04_import_effect.txt).
    FastPolicy p{W, Bp, j, dp.slot_of};
    uint32_t *sc = nullptr;
    if (scratch) {
        sc = scratch + (uint64_t)scratch_off[2u * blockIdx.y] * Bp + (j >> 6) * 64u * (uint64_t)scratch_off[2u * blockIdx.y + 1u];
        p.sBp = 64u;
        p.sj = j & 63u;
    }
    const OpResult r = Op::run(p, rec, dp, sc, (SlowResult *)nullptr, (const ExactLanes *)nullptr, 0u);
    if (r.err) flag_instance(event, j, rec[0] == K_RANGE_MULTI ? r.aux0 : rec[1]);  // (a merged record names the failing opcode itself)
}

template <class Op, int BLOCK>
static void launch_record_level(hipStream_t s, uint4 *W, uint64_t Bp, uint32_t B, const DeviceProgram &dp, const uint32_t *offsets,
                                const uint32_t *scratch_off, uint32_t n, uint32_t *event, uint32_t *scratch) {
    if (!n || !B) return;
    for (uint32_t done = 0; done < n;) {  // gridDim.y is limited to 65535
        const uint32_t m = n - done > 65535u ? 65535u : n - done;
        hipLaunchKernelGGL((record_level_kernel<Op, BLOCK>), dim3((B + BLOCK - 1) / BLOCK, m), dim3(BLOCK), 0, s, W, Bp, B, dp, offsets + done,
                           scratch_off ? scratch_off + 2 * (size_t)done : nullptr, event, scratch, dp.prog, dp.consts, dp.slot_of, dp.bytecode);
        done += m;
    }
}
}  // namespace acvm
