// kernels_hash.hip -- SHA256 / Blake2s / Keccak256(+variable length) / HashToField128Security opcodes
// (acvm/src/pwg/blackbox/hash.rs; device routines in ops_hash.hpp), level kernel + exact kernel. The same scratch-carrying
// kernels also run Directive::PermutationSort (ops_sort.hpp), the other opcode that needs per-lane working memory.
#include "ops_hash.hpp"
#include "ops_sort.hpp"
#include "ops_kernel.hpp"

namespace acvm {

struct HashOp {
    template <class P>
    static __device__ __forceinline__ OpResult run(const P &p, const uint32_t *__restrict__ rec, const DeviceProgram &dp, uint32_t *scratch, SlowResult *, const ExactLanes *, uint32_t) {
        if (rec[0] == K_PERM_SORT) return op_perm_sort(p, rec, dp.consts, scratch);
        return op_hash(p, rec, scratch);
    }
};

void launch_hash_level(hipStream_t s, uint4 *W, uint64_t Bp, uint32_t B, const DeviceProgram &dp, const uint32_t *offsets,
                       const uint32_t *scratch_off, uint32_t n, uint32_t *event, uint32_t *scratch) {
    launch_record_level<HashOp, 128>(s, W, Bp, B, dp, offsets, scratch_off, n, event, scratch);
}
void launch_exact_hash(hipStream_t s, uint4 *W, uint64_t Bp, const DeviceProgram &dp, const ExactLanes &L, uint32_t opcode, uint32_t *scratch) {
    launch_record_exact<HashOp, 64>(s, W, Bp, dp, L, opcode, scratch);
}

}  // namespace acvm
