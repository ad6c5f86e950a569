// kernels_ecdsa.hip -- EcdsaSecp256k1 / EcdsaSecp256r1 opcodes (device routines in ops_ecdsa.hpp), level + exact kernel.
#include "ops_ecdsa.hpp"
#include "ops_kernel.hpp"

namespace acvm {

struct EcdsaOp {
    template <class P>
    static __device__ __forceinline__ OpResult run(const P &p, const uint32_t *__restrict__ rec, const DeviceProgram &, uint32_t *, SlowResult *, const ExactLanes *, uint32_t) {
        return op_ecdsa(p, rec);
    }
};

void launch_ecdsa_level(hipStream_t s, uint4 *W, uint64_t Bp, uint32_t B, const DeviceProgram &dp, const uint32_t *offsets, uint32_t n, uint32_t *event) {
    launch_record_level<EcdsaOp, 64>(s, W, Bp, B, dp, offsets, nullptr, n, event, nullptr);
}

}  // namespace acvm
