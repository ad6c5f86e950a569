// kernels_brillig.hip -- Opcode::Brillig: the Brillig VM (ops_brillig.hpp) as a level kernel and as an exact kernel.
#include "ops_brillig.hpp"
#include "ops_kernel.hpp"
#include "ops_sort.hpp"

namespace acvm {

struct BrilligOp {
    template <class P>
    static __device__ __forceinline__ OpResult run(const P &p, const uint32_t *__restrict__ rec, const DeviceProgram &dp, uint32_t *scratch, SlowResult *res, const ExactLanes *L, uint32_t t) {
        return op_brillig(p, rec, dp, scratch, res, L, t);
    }
};

void launch_brillig_level(hipStream_t s, uint4 *W, uint64_t Bp, uint32_t B, const DeviceProgram &dp, const uint32_t *offsets,
                          const uint32_t *scratch_off, uint32_t n, uint32_t *event, uint32_t *scratch) {
    launch_record_level<BrilligOp, 64>(s, W, Bp, B, dp, offsets, scratch_off, n, event, scratch);
}

// ---------------------------------------------------------------------------------------------- the exact path in one launch
// Opcodes [op_begin, op_end) in program order for the lanes of the exact path (acvm/src/pwg/mod.rs:236-303 for the instances that left
// the generic path), EVERY opcode class in this one kernel: a lane per flagged instance walks the in-order program and dispatches on the
// opcode's class. Round 2 launched one kernel per heavy opcode and one per span of light opcodes: on the 10^6-opcode circuit that is
// ~20 000 launches of one wave each, 3 us apiece -- five diverging instances cost a tile +64 ms. Occupancy is irrelevant here (a handful
// of lanes), so the kernel may be as fat as its fattest class. Opcodes before a lane's start only replay their memory side effects
// (a later opcode of the level schedule may already have overwritten the cell; their witness outputs are kept: init_assigned_kernel).
// A caller-supplied BlackBoxFunctionSolver still splits the run at its opcodes (batch.cpp).
__global__ void __launch_bounds__(64) exact_run_kernel(uint4 *W, uint64_t Bp, DeviceProgram dp, ExactLanes L, uint32_t op_begin, uint32_t op_end, uint32_t replay_memory,
                                                       const uint8_t *__restrict__ prog_class, ExactScratch sc) {
    const uint32_t t = blockIdx.x * 64 + threadIdx.x;
    if (t >= L.n_slow) return;
    if (L.results[t].status != 1u) return;
    const uint64_t j = L.slow_ids[t];
    const uint32_t start = L.start_opcode[t];
    ExactPolicy p{W, Bp, j, L.assigned, L.n_slow, t};
    FastPolicy replay{W, Bp, j, nullptr};  // the exact path addresses its table by witness index
    for (uint32_t oi = replay_memory || start < op_begin ? op_begin : start; oi < op_end; oi++) {
        const uint32_t *__restrict__ rec = dp.prog + dp.prog_offset[oi];
        if (oi < start) {
            if (rec[0] == K_MEM_INIT) op_mem_init(replay, rec, dp.Mem);
            else if (rec[0] == K_MEM_OP) op_mem_op(replay, rec, dp.consts, dp.Mem, true);
            continue;
        }
        OpResult r;
        switch (prog_class[oi]) {
        case 0: r = dispatch_light(p, rec, dp.consts, dp.Mem); break;                                                      // CLS_LIGHT
        case 1: r = rec[0] == K_PERM_SORT ? op_perm_sort(p, rec, dp.consts, sc.hash) : op_hash(p, rec, sc.hash); break;    // CLS_HASH
        case 2: r = dispatch_grumpkin(p, rec, dp.grumpkin, sc.grumpkin); break;                                            // CLS_GRUMPKIN (Pedersen included)
        case 3: r = op_brillig(p, rec, dp, sc.brillig, &L.results[t], &L, t); break;                                       // CLS_BRILLIG
        case 6: r = op_ecdsa(p, rec, dp.ecdsa_g); break;                                                                               // CLS_ECDSA
        default: r = op_fail_msg(DE_PANIC, 0, DM_NONE); break;  // (CLS_HOSTBB never reaches this kernel)
        }
        if (r.err == DE_WAIT_FOREIGN_CALL) {  // ACVMStatus::RequiresForeignCall: the instruction pointer stays on this opcode
            L.results[t].status = 3u;
            L.results[t].opcode_index = oi;
            L.results[t].x0 = r.x0;
            return;
        }
        if (r.err) { exact_fail(L, t, oi, r); return; }
    }
}
void launch_exact_run(hipStream_t s, uint4 *W, uint64_t Bp, const DeviceProgram &dp, const ExactLanes &L, uint32_t op_begin, uint32_t op_end, bool replay_memory,
                      const uint8_t *prog_class, const ExactScratch &sc) {
    if (!L.n_slow || op_begin >= op_end) return;
    hipLaunchKernelGGL(exact_run_kernel, dim3((L.n_slow + 63) / 64), dim3(64), 0, s, W, Bp, dp, L, op_begin, op_end, replay_memory ? 1u : 0u, prog_class, sc);
}

}  // namespace acvm
