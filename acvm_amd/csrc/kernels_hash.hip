// kernels_hash.hip -- SHA256 / Blake2s / Keccak256(+variable length) / HashToField128Security opcodes
// (acvm/src/pwg/blackbox/hash.rs; device routines in ops_hash.hpp), level kernel + exact kernel. The same scratch-carrying
// kernels also run Directive::PermutationSort (ops_sort.hpp), the other opcode that needs per-lane working memory.
#include <algorithm>
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

// Level kernel of the records flagged HASH_COOP_FLAG (byte messages of SHA256 / Blake2s / Keccak256; batch.cpp launches the two kinds of a
// level separately). At the batch sizes of one tile a SIMD holds ONE wave of a lane-per-instance kernel, and a byte message costs that lane
// one table row and one reduction per BYTE before the first compression: latency-bound on those rows (VALU 29 % busy). Here a block of four
// waves serves 64 instances of one record: wave q fetches and reduces the bytes q, q + 4, q + 8, ... of every instance (four rows in flight per
// lane, four times the waves in flight per SIMD) into the LDS message, wave 0 hashes, and the 32 digest bytes go back through LDS so that
// wave q stores outputs 8q .. 8q + 7 (WAVES = 4). The hash bodies are inlined here under the kernel's register budget (four waves per SIMD = 128 VGPRs).
// the RANGE opcode fused on input i of the record fails: its opcode index, else 0xFFFFFFFF (is_byte: the value is the byte `low`)
__device__ __forceinline__ uint32_t range_check(const uint32_t *__restrict__ ranges, uint32_t i, bool is_byte, uint32_t low) {
    const uint32_t op = ranges[2u * i], bits = ranges[2u * i + 1u];
    if (op == 0xFFFFFFFFu) return op;
    return is_byte && (low >> bits) == 0u ? 0xFFFFFFFFu : op;
}
// WAVES = 1 is the same kernel for launches that fill the chip anyhow (many records per level): one wave per 64 instances does every
// phase, still with the message in LDS and the lean register budget (the lane-per-instance kernel with the message in device scratch
// and 184 VGPRs measured 6.0 ms for the hash class of the config-5 mix at 2^16 instances).
template <int WAVES>
__global__ void __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(4, 8)))
hash_coop_level_kernel(uint4 *W, uint64_t Bp, uint32_t B, DeviceProgram dp, const uint32_t *__restrict__ offsets, uint32_t *__restrict__ event,
                       const uint32_t *__restrict__ prog, const uint32_t *__restrict__ slot_of) {
    extern __shared__ uint32_t lds[];  // max(message words of the launch's longest record, 8) x 64 words
    // (the wave index as a scalar: everything indexed by it -- record words, witness ids, rows of slot_of -- is then a scalar load; as a
    // vector value each of those was a memory round trip of its own in front of every row)
    const uint32_t lane = threadIdx.x & 63u, q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t j = (uint64_t)blockIdx.x * 64u + lane;
    // (prog and slot_of arrive as kernel arguments of their own, not inside dp: only a noalias argument lets the compiler read the record
    // with scalar loads; through the struct every record word was a vector load that also waited for the stores before it)
    const uint32_t *__restrict__ rec = prog + offsets[blockIdx.y];
    FastPolicy p{W, Bp, j, slot_of};
    const uint32_t func = rec[2] & 0xffu, n_in = rec[3];
    const uint32_t *ins = rec + 6, *outs = ins + 2 * n_in;
    const bool live = j < B;  // (rows are padded to Bp: the loads of a dead lane stay inside the table)
    uint8_t *bytes = (uint8_t *)lds;
    const uint32_t *ranges = (rec[2] & HASH_RANGE_FLAG) ? outs + 64 : nullptr;  // (opcode or NONE, bits) per input
    uint32_t range_bad = 0xFFFFFFFFu;
    if (q == 0 && (n_in & 3u)) lds[(n_in >> 2) * 64u + lane] = 0u;  // the bytes behind the message in its last word
    __syncthreads();
    for (uint32_t i = q; i < n_in; i += 4u * WAVES) {  // wave-uniform bounds
        const uint32_t i1 = i + WAVES, i2 = i + 2u * WAVES, i3 = i + 3u * WAVES;
        const Fr a0 = p.load(ins[2 * i]);
        const Fr a1 = i1 < n_in ? p.load(ins[2 * i1]) : a0, a2 = i2 < n_in ? p.load(ins[2 * i2]) : a0, a3 = i3 < n_in ? p.load(ins[2 * i3]) : a0;
        bool b0, b1, b2, b3;  // the value is a byte (then l is that byte)
        const uint32_t l0 = fr_low_limb(a0, b0), l1 = fr_low_limb(a1, b1), l2 = fr_low_limb(a2, b2), l3 = fr_low_limb(a3, b3);
        bytes[4u * ((i >> 2) * 64u + lane) + (i & 3u)] = (uint8_t)l0;
        if (i1 < n_in) bytes[4u * ((i1 >> 2) * 64u + lane) + (i1 & 3u)] = (uint8_t)l1;
        if (i2 < n_in) bytes[4u * ((i2 >> 2) * 64u + lane) + (i2 & 3u)] = (uint8_t)l2;
        if (i3 < n_in) bytes[4u * ((i3 >> 2) * 64u + lane) + (i3 & 3u)] = (uint8_t)l3;
        if (ranges) {  // the RANGE opcodes fused into this record (plan.cpp): same test as op_range on the limb that is here already
            range_bad = min(range_bad, range_check(ranges, i, b0, l0));
            if (i1 < n_in) range_bad = min(range_bad, range_check(ranges, i1, b1, l1));
            if (i2 < n_in) range_bad = min(range_bad, range_check(ranges, i2, b2, l2));
            if (i3 < n_in) range_bad = min(range_bad, range_check(ranges, i3, b3, l3));
        }
    }
    if (range_bad != 0xFFFFFFFFu && live) atomicMin(&event[j], range_bad);
    __syncthreads();
    Digest d;
    if (q == 0) {
        const LdsMsg m{lds, lane};
        if (func == 3u) d = sha256_body(m, n_in);
        else if (func == 4u) d = blake2s_body(m, n_in);
        else d = keccak256_body(m, n_in);
    }
    __syncthreads();  // the message has been read
    if (q == 0) {
#pragma unroll
        for (int k = 0; k < 8; k++) lds[(uint32_t)k * 64u + lane] = d.d[k];
    }
    __syncthreads();
    if (!live) return;
    bool ok = true;
    for (uint32_t k = 0; k < 32u / WAVES; k++) {
        const uint32_t i = (32u / WAVES) * q + k;
        const uint32_t byte = (lds[(i >> 2) * 64u + lane] >> (8u * (i & 3u))) & 0xffu;
        ok = p.insert(outs[2 * i], fr_from_byte(byte), outs[2 * i + 1]) && ok;  // (hash.rs:92-103 stops at the first conflict; the flagged instance re-runs exactly)
    }
    if (!ok) atomicMin(&event[j], rec[1]);
}
void launch_hash_coop_level(hipStream_t s, uint4 *W, uint64_t Bp, uint32_t B, const DeviceProgram &dp, const uint32_t *offsets, uint32_t n, uint32_t *event, uint32_t lds_words) {
    if (!n || !B) return;
    const size_t lds_bytes = (size_t)std::max<uint32_t>(lds_words, 8u) * 64u * 4u;
    const uint64_t groups = (uint64_t)((B + 63u) / 64u) * n;  // one per 64 instances of a record
    const bool four = groups * 4u <= 8192u;                    // four waves each while that still fits the chip about twice (1 024 SIMDs x 4-5 waves)
    for (uint32_t done = 0; done < n;) {  // gridDim.y is limited to 65535
        const uint32_t m = n - done > 65535u ? 65535u : n - done;
        if (four) hipLaunchKernelGGL(hash_coop_level_kernel<4>, dim3((B + 63u) / 64u, m), dim3(256), lds_bytes, s, W, Bp, B, dp, offsets + done, event, dp.prog, dp.slot_of);
        else hipLaunchKernelGGL(hash_coop_level_kernel<1>, dim3((B + 63u) / 64u, m), dim3(64), lds_bytes, s, W, Bp, B, dp, offsets + done, event, dp.prog, dp.slot_of);
        done += m;
    }
}
void launch_exact_hash(hipStream_t s, uint4 *W, uint64_t Bp, const DeviceProgram &dp, const ExactLanes &L, uint32_t opcode, uint32_t *scratch) {
    launch_record_exact<HashOp, 64>(s, W, Bp, dp, L, opcode, scratch);
}

// ------------------------------------------------------------------------------------------ witness-map digest
// SURVEY 8d (config 5): callers that do not want the full map back keep the return witnesses and a 32-byte digest per instance.
// Definition (include/acvm_amd.h acvm_batch_digest): witnesses 2i and 2i + 1 form pair i; mask = 1 (2i assigned) | 2 (2i + 1 assigned);
// a pair with mask != 0 has the leaf Blake2s-256(message = the 32-byte big-endian canonical values of its assigned witnesses in ascending
// order, personalisation = le32(i) || le32(mask)) -- ONE compression; S = the sum of all leaves taken as eight little-endian 32-bit words,
// each word modulo 2^32; digest = Blake2s-256(S). The sum makes the leaves order-free: a leaf is hashed as soon as its two witnesses
// exist (the folded digest of plan.cpp, which lets witness rows be recycled early), by any number of lanes, and added with atomics.
__device__ __forceinline__ void digest_leaf_add(uint32_t (&S)[8], uint32_t pair, uint32_t mask, const Fr &ca, const Fr &cb) {
    uint32_t h[8], w[16];
    blake2s_init(h);
    h[6] ^= pair;  // parameter block bytes 24..31: the personalisation
    h[7] ^= mask;
    const Fr &first = (mask & 1u) ? ca : cb;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        w[k] = bswap32(first.v[7 - k]);  // big-endian bytes as little-endian message words
        w[8 + k] = mask == 3u ? bswap32(cb.v[7 - k]) : 0u;
    }
    blake2s_compress_body(h, w, mask == 3u ? 64u : 32u, true);
#pragma unroll
    for (int k = 0; k < 8; k++) S[k] += h[k];
}
static constexpr uint32_t DIGEST_CHUNK_PAIRS = 64;  // pairs one lane of the post-solve kernel hashes
// after the solve, from the table: lane = (instance, chunk of pairs); acc = [8][n] words, zeroed by the launcher
__global__ void __launch_bounds__(128) digest_pairs_kernel(const uint4 *__restrict__ W, uint64_t Bp, uint32_t first, uint32_t n, uint32_t n_witnesses,
                                                           const uint32_t *__restrict__ producer, const Unscale u, const int32_t *__restrict__ slow_index,
                                                           const uint32_t *__restrict__ assigned, uint32_t n_slow, uint32_t *__restrict__ acc, uint32_t chunk0) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, chunk = chunk0 + blockIdx.y;
    if (t >= n) return;
    const uint64_t j = (uint64_t)first + t;
    const bool generic = !u.event || u.event[j] == 0xFFFFFFFFu;  // solved by the level kernels: the planner's assigned set, scaled columns (no event words: every lane)
    const uint32_t lane = generic ? 0u : (uint32_t)slow_index[j];
    Fr one = fr_zero();
    one.v[0] = 1;
    uint32_t S[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const uint32_t p0 = chunk * DIGEST_CHUNK_PAIRS, n_pairs = (n_witnesses + 1) / 2;
    for (uint32_t pi = p0; pi < p0 + DIGEST_CHUNK_PAIRS && pi < n_pairs; pi++) {
        uint32_t mask = 0;
        Fr c[2] = {fr_zero(), fr_zero()};
        for (uint32_t h = 0; h < 2; h++) {
            const uint32_t w = 2 * pi + h;
            if (w >= n_witnesses) continue;
            const bool present = generic ? producer[w] != 0xFFFFFFFFu : ((assigned[(uint64_t)(w >> 5) * n_slow + lane] >> (w & 31)) & 1u) != 0u;
            if (!present) continue;
            mask |= 1u << h;
            const uint32_t ui = generic && u.index ? u.index[w] : 0xFFFFFFFFu;
            c[h] = fr_mul(fr_load(W, w, Bp, j), ui != 0xFFFFFFFFu ? fr_const(u.consts_plain, ui) : one);  // canonical value (unscaled where the column is scaled)
        }
        if (mask) digest_leaf_add(S, pi, mask, c[0], c[1]);
    }
#pragma unroll
    for (int k = 0; k < 8; k++)
        if (S[k]) atomicAdd(&acc[(uint64_t)k * n + t], S[k]);
}
// The same leaves DURING the solve (PlanOpts::fold_digest): records [PK_DIGEST_LEAF, 0, n, (pair, witness 2i or NONE, witness 2i + 1 or NONE,
// row of unscale or NONE x 2) x n] of the pairs whose witnesses are complete (the planner's assigned set = the generic instance), read through
// the row map of slot reuse; acc = [8][Bp]. Instances that leave the generic path are hashed from the exact path's map afterwards.
__global__ void __launch_bounds__(128) digest_fold_level_kernel(const uint4 *__restrict__ W, uint64_t Bp, uint32_t B, const uint32_t *__restrict__ prog,
                                                                const uint32_t *__restrict__ offsets, const uint32_t *__restrict__ slot_of,
                                                                const uint32_t *__restrict__ unscale_plain, uint32_t *__restrict__ acc) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= B) return;
    const uint32_t *__restrict__ r = prog + offsets[blockIdx.y];
    const uint32_t n = r[2];
    Fr one = fr_zero();
    one.v[0] = 1;
    uint32_t S[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t *__restrict__ e = r + 3 + 5 * i;
        uint32_t mask = 0;
        Fr c[2] = {fr_zero(), fr_zero()};
        for (uint32_t h = 0; h < 2; h++) {
            const uint32_t w = e[1 + h], ui = e[3 + h];
            if (w == 0xFFFFFFFFu) continue;
            mask |= 1u << h;
            c[h] = fr_mul(fr_load(W, slot_of ? slot_of[w] : w, Bp, j), ui != 0xFFFFFFFFu ? fr_const(unscale_plain, ui) : one);
        }
        digest_leaf_add(S, e[0], mask, c[0], c[1]);
    }
#pragma unroll
    for (int k = 0; k < 8; k++)
        if (S[k]) atomicAdd(&acc[(uint64_t)k * Bp + j], S[k]);
}
// digest = Blake2s-256(S); acc laid out [8][stride], instances [first, first + n)
__global__ void __launch_bounds__(128) digest_final_kernel(const uint32_t *__restrict__ acc, uint64_t stride, uint32_t first, uint32_t n, uint8_t *__restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    uint32_t h[8], w[16];
    blake2s_init(h);
#pragma unroll
    for (int k = 0; k < 8; k++) { w[k] = acc[(uint64_t)k * stride + first + t]; w[8 + k] = 0u; }
    blake2s_compress(h, w, 32u, true);
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int k = 0; k < 4; k++) out[(uint64_t)t * 32 + 4 * i + k] = (uint8_t)(h[i] >> (8 * k));
}
__global__ void zero_u32_kernel(uint32_t *p, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0u;
}
// acc: scratch of 8 x n words
void launch_digest(hipStream_t s, const uint4 *W, uint64_t Bp, uint32_t first, uint32_t n, uint32_t n_witnesses, const uint32_t *producer, const Unscale &u,
                   const int32_t *slow_index, const uint32_t *assigned, uint32_t n_slow, uint32_t *acc, uint8_t *out) {
    if (!n) return;
    hipLaunchKernelGGL(zero_u32_kernel, dim3((unsigned)((8ull * n + 255) / 256)), dim3(256), 0, s, acc, 8ull * n);
    const uint32_t n_chunks = ((n_witnesses + 1) / 2 + DIGEST_CHUNK_PAIRS - 1) / DIGEST_CHUNK_PAIRS;
    for (uint32_t done = 0; done < n_chunks; done += 65535u) {  // gridDim.y is limited to 65535
        const uint32_t m = n_chunks - done > 65535u ? 65535u : n_chunks - done;
        hipLaunchKernelGGL(digest_pairs_kernel, dim3((n + 127) / 128, m), dim3(128), 0, s, W, Bp, first, n, n_witnesses, producer, u, slow_index, assigned, n_slow,
                           acc, done);
    }
    hipLaunchKernelGGL(digest_final_kernel, dim3((n + 127) / 128), dim3(128), 0, s, acc, (uint64_t)n, 0u, n, out);
}
void launch_digest_fold_level(hipStream_t s, const uint4 *W, uint64_t Bp, uint32_t B, const DeviceProgram &dp, const uint32_t *offsets, uint32_t n,
                              const uint32_t *unscale_plain, uint32_t *acc) {
    if (!n || !B) return;
    for (uint32_t done = 0; done < n; done += 65535u) {
        const uint32_t m = n - done > 65535u ? 65535u : n - done;
        hipLaunchKernelGGL(digest_fold_level_kernel, dim3((B + 127) / 128, m), dim3(128), 0, s, W, Bp, B, dp.prog, offsets + done, dp.slot_of, unscale_plain, acc);
    }
}
void launch_digest_final(hipStream_t s, const uint32_t *acc, uint64_t stride, uint32_t first, uint32_t n, uint8_t *out) {
    if (!n) return;
    hipLaunchKernelGGL(digest_final_kernel, dim3((n + 127) / 128), dim3(128), 0, s, acc, stride, first, n, out);
}

}  // namespace acvm
