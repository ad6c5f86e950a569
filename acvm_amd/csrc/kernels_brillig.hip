// kernels_brillig.hip -- Opcode::Brillig: the Brillig VM (ops_brillig.hpp) as a level kernel and as an exact kernel.
#include "ops_brillig.hpp"
#include "ops_kernel.hpp"

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
void launch_exact_brillig(hipStream_t s, uint4 *W, uint64_t Bp, const DeviceProgram &dp, const ExactLanes &L, uint32_t opcode, uint32_t *scratch) {
    launch_record_exact<BrilligOp, 64>(s, W, Bp, dp, L, opcode, scratch);
}

}  // namespace acvm
