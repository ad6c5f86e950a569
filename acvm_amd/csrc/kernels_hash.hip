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

// ------------------------------------------------------------------------------------------ witness-map digest
// SURVEY 8d (config 5): callers that do not want the full map back keep the return witnesses and a 32-byte digest per instance.
// Definition (include/acvm_amd.h acvm_batch_digest): the assigned witnesses in ascending index, each as its 32-byte big-endian
// canonical value, are cut into segments of DIGEST_SEG witness INDICES; leaf_k = Blake2s-256 of the assigned ones among
// [k DIGEST_SEG, (k+1) DIGEST_SEG); digest = Blake2s-256(leaf_0 || leaf_1 || ...). Two levels so that (segments x instances)
// lanes hash concurrently: a single Blake2s chain per instance would leave a 4 096-instance tile of a 10^6-witness circuit on 64 waves.
static constexpr uint32_t DIGEST_SEG = 256;
__global__ void __launch_bounds__(128) digest_leaf_kernel(const uint4 *__restrict__ W, uint64_t Bp, uint32_t first, uint32_t n, uint32_t n_witnesses,
                                                          const uint32_t *__restrict__ producer, const Unscale u, const int32_t *__restrict__ slow_index,
                                                          const uint32_t *__restrict__ assigned, uint32_t n_slow, uint32_t *__restrict__ leaves, uint32_t seg0) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, seg = seg0 + blockIdx.y;
    if (t >= n) return;
    const uint64_t j = (uint64_t)first + t;
    const bool generic = u.event[j] == 0xFFFFFFFFu;  // solved by the level kernels: the planner's assigned set, scaled columns
    const uint32_t lane = generic ? 0u : (uint32_t)slow_index[j];
    Blake2sPieces st;
    st.begin();
    const uint32_t w0 = seg * DIGEST_SEG, w1 = w0 + DIGEST_SEG < n_witnesses ? w0 + DIGEST_SEG : n_witnesses;
    for (uint32_t w = w0; w < w1; w++) {
        const bool present = generic ? producer[w] != 0xFFFFFFFFu : ((assigned[(uint64_t)(w >> 5) * n_slow + lane] >> (w & 31)) & 1u) != 0u;
        if (!present) continue;
        Fr x = fr_load(W, w, Bp, j);
        const uint32_t ui = generic && u.index ? u.index[w] : 0xFFFFFFFFu;  // wave-uniform unless the wave holds flagged instances
        Fr one = fr_zero();
        one.v[0] = 1;
        x = fr_mul(x, ui != 0xFFFFFFFFu ? fr_const(u.consts_plain, ui) : one);  // canonical value (unscaled where the column is scaled)
        uint32_t m[8];
#pragma unroll
        for (int i = 0; i < 8; i++) m[i] = bswap32(x.v[7 - i]);  // big-endian bytes as little-endian message words
        st.put(m);
    }
    uint32_t d[8];
    st.finish(d);
#pragma unroll
    for (int i = 0; i < 8; i++) leaves[((uint64_t)seg * 8 + i) * n + t] = d[i];
}
__global__ void __launch_bounds__(128) digest_root_kernel(const uint32_t *__restrict__ leaves, uint32_t n, uint32_t n_seg, uint8_t *__restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    Blake2sPieces st;
    st.begin();
    for (uint32_t seg = 0; seg < n_seg; seg++) {
        uint32_t m[8];
#pragma unroll
        for (int i = 0; i < 8; i++) m[i] = leaves[((uint64_t)seg * 8 + i) * n + t];
        st.put(m);
    }
    uint32_t d[8];
    st.finish(d);
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int k = 0; k < 4; k++) out[(uint64_t)t * 32 + 4 * i + k] = (uint8_t)(d[i] >> (8 * k));
}
// The same leaf DURING the solve (PlanOpts::fold_digest): one record per segment ([PK_DIGEST_LEAF, segment, n, (witness, row of unscale
// or NONE) x n], the planner's assigned set = the generic instance), scheduled right behind the last witness of the segment. Reads the
// table through the row map of slot reuse. Instances that leave the generic path get their leaves from the exact path's map afterwards.
__global__ void __launch_bounds__(128) digest_fold_level_kernel(const uint4 *__restrict__ W, uint64_t Bp, uint32_t B, const uint32_t *__restrict__ prog,
                                                                const uint32_t *__restrict__ offsets, const uint32_t *__restrict__ slot_of,
                                                                const uint32_t *__restrict__ unscale_plain, uint32_t *__restrict__ leaves) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= B) return;
    const uint32_t *__restrict__ r = prog + offsets[blockIdx.y];
    const uint32_t seg = r[1], n = r[2];
    Blake2sPieces st;
    st.begin();
    Fr one = fr_zero();
    one.v[0] = 1;
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t w = r[3 + 2 * i], ui = r[4 + 2 * i];
        Fr x = fr_load(W, slot_of ? slot_of[w] : w, Bp, j);
        x = fr_mul(x, ui != 0xFFFFFFFFu ? fr_const(unscale_plain, ui) : one);  // canonical value (unscaled where the column is scaled)
        uint32_t m[8];
#pragma unroll
        for (int k = 0; k < 8; k++) m[k] = bswap32(x.v[7 - k]);
        st.put(m);
    }
    uint32_t d[8];
    st.finish(d);
#pragma unroll
    for (int k = 0; k < 8; k++) leaves[((uint64_t)seg * 8 + k) * Bp + j] = d[k];
}
void launch_digest_fold_level(hipStream_t s, const uint4 *W, uint64_t Bp, uint32_t B, const DeviceProgram &dp, const uint32_t *offsets, uint32_t n,
                              const uint32_t *unscale_plain, uint32_t *leaves) {
    if (!n || !B) return;
    for (uint32_t done = 0; done < n; done += 65535u) {
        const uint32_t m = n - done > 65535u ? 65535u : n - done;
        hipLaunchKernelGGL(digest_fold_level_kernel, dim3((B + 127) / 128, m), dim3(128), 0, s, W, Bp, B, dp.prog, offsets + done, dp.slot_of, unscale_plain, leaves);
    }
}
// the root over leaves laid out [seg][8][stride] (the folded digest keeps them at the batch's stride)
void launch_digest_root(hipStream_t s, const uint32_t *leaves, uint64_t stride, uint32_t first, uint32_t n, uint32_t n_seg, uint8_t *out);
__global__ void __launch_bounds__(128) digest_root_strided_kernel(const uint32_t *__restrict__ leaves, uint64_t stride, uint32_t first, uint32_t n, uint32_t n_seg,
                                                                  uint8_t *__restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    Blake2sPieces st;
    st.begin();
    for (uint32_t seg = 0; seg < n_seg; seg++) {
        uint32_t m[8];
#pragma unroll
        for (int i = 0; i < 8; i++) m[i] = leaves[((uint64_t)seg * 8 + i) * stride + first + t];
        st.put(m);
    }
    uint32_t d[8];
    st.finish(d);
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int k = 0; k < 4; k++) out[(uint64_t)t * 32 + 4 * i + k] = (uint8_t)(d[i] >> (8 * k));
}
void launch_digest_root(hipStream_t s, const uint32_t *leaves, uint64_t stride, uint32_t first, uint32_t n, uint32_t n_seg, uint8_t *out) {
    if (!n) return;
    hipLaunchKernelGGL(digest_root_strided_kernel, dim3((n + 127) / 128), dim3(128), 0, s, leaves, stride, first, n, n_seg, out);
}
uint32_t digest_segments(uint32_t n_witnesses) { return (n_witnesses + DIGEST_SEG - 1) / DIGEST_SEG; }
void launch_digest(hipStream_t s, const uint4 *W, uint64_t Bp, uint32_t first, uint32_t n, uint32_t n_witnesses, const uint32_t *producer, const Unscale &u,
                   const int32_t *slow_index, const uint32_t *assigned, uint32_t n_slow, uint32_t *leaves, uint8_t *out) {
    if (!n) return;
    const uint32_t n_seg = digest_segments(n_witnesses);
    for (uint32_t done = 0; done < n_seg; done += 65535u) {  // gridDim.y is limited to 65535
        const uint32_t m = n_seg - done > 65535u ? 65535u : n_seg - done;
        hipLaunchKernelGGL(digest_leaf_kernel, dim3((n + 127) / 128, m), dim3(128), 0, s, W, Bp, first, n, n_witnesses, producer, u, slow_index, assigned, n_slow,
                           leaves, done);
    }
    hipLaunchKernelGGL(digest_root_kernel, dim3((n + 127) / 128), dim3(128), 0, s, leaves, n, n_seg, out);
}

}  // namespace acvm
