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
// Chains (plan.cpp "hash chains"): a record whose function word carries HASH_CHAIN_FLAG is followed, in the same block, by the byte-message hash that
// consumes its digest (a hash of a hash, a Merkle path): the word behind the record is the offset of a link [offset of the next record, source of
// each of its inputs: index of the previous digest's byte or NONE]; the next record takes those bytes from LDS instead of reading back the rows the
// block has just written, and the launch of its own (and the lock-step read / hash / write phases of that launch) is gone.
template <int WAVES>
__global__ void __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(4, 8)))
hash_coop_level_kernel(uint4 *W, uint64_t Bp, uint32_t B, DeviceProgram dp, const uint32_t *__restrict__ offsets, uint32_t *__restrict__ event,
                       const uint32_t *__restrict__ prog, const uint32_t *__restrict__ slot_of) {
    extern __shared__ uint32_t lds[];       // max(message words of the launch's longest record, 8) x 64 words
    __shared__ uint32_t lds_prev[8][64];    // the digest of the previous record of a chain
    // (the wave index as a scalar: everything indexed by it -- record words, witness ids, rows of slot_of -- is then a scalar load; as a
    // vector value each of those was a memory round trip of its own in front of every row)
    const uint32_t lane = threadIdx.x & 63u, q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t j = (uint64_t)blockIdx.x * 64u + lane;
    // (prog and slot_of arrive as kernel arguments of their own, not inside dp: only a noalias argument lets the compiler read the record
    // with scalar loads; through the struct every record word was a vector load that also waited for the stores before it)
    const uint32_t *__restrict__ rec = prog + offsets[blockIdx.y];
    const uint32_t *__restrict__ src = nullptr;  // per input of a chained record: byte of the previous digest, or NONE (null: the head of a chain)
    FastPolicy p{W, Bp, j, slot_of};
    const bool live = j < B;  // (rows are padded to Bp: the loads of a dead lane stay inside the table)
    uint8_t *bytes = (uint8_t *)lds;
    for (;;) {
        const uint32_t func = rec[2] & 0xffu, n_in = rec[3];
        const uint32_t *ins = rec + 6, *outs = ins + 2 * n_in;
        const uint32_t *ranges = (rec[2] & HASH_RANGE_FLAG) ? outs + 64 : nullptr;  // (opcode or NONE, bits) per input
        uint32_t range_bad = 0xFFFFFFFFu;
        if (q == 0 && (n_in & 3u)) lds[(n_in >> 2) * 64u + lane] = 0u;  // the bytes behind the message in its last word
        __syncthreads();
        // one input: (low limb, is a byte) from its row, or the byte of the previous digest
        auto fetch = [&](uint32_t i, bool &is_byte) {
            const uint32_t from = src ? src[i] : 0xFFFFFFFFu;  // (scalar: i is wave-uniform)
            if (from != 0xFFFFFFFFu) {
                is_byte = true;
                return (lds_prev[from >> 2][lane] >> (8u * (from & 3u))) & 0xffu;
            }
            return fr_low_limb(p.load(ins[2 * i]), is_byte);
        };
        for (uint32_t i = q; i < n_in; i += 4u * WAVES) {  // wave-uniform bounds
            const uint32_t i1 = i + WAVES, i2 = i + 2u * WAVES, i3 = i + 3u * WAVES;
            bool b0, b1 = true, b2 = true, b3 = true;  // the value is a byte (then l is that byte)
            const uint32_t l0 = fetch(i, b0), l1 = i1 < n_in ? fetch(i1, b1) : 0u, l2 = i2 < n_in ? fetch(i2, b2) : 0u, l3 = i3 < n_in ? fetch(i3, b3) : 0u;
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
        const bool more = (rec[2] & HASH_CHAIN_FLAG) != 0u;
        if (q == 0) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                lds[(uint32_t)k * 64u + lane] = d.d[k];
                if (more) lds_prev[k][lane] = d.d[k];
            }
        }
        __syncthreads();
        if (live) {
            bool ok = true;
            for (uint32_t k = 0; k < 32u / WAVES; k++) {
                const uint32_t i = (32u / WAVES) * q + k;
                const uint32_t byte = (lds[(i >> 2) * 64u + lane] >> (8u * (i & 3u))) & 0xffu;
                ok = p.insert(outs[2 * i], fr_from_byte(byte), outs[2 * i + 1]) && ok;  // (hash.rs:92-103 stops at the first conflict; the flagged instance re-runs exactly)
            }
            if (!ok) atomicMin(&event[j], rec[1]);
        }
        if (!more) break;
        const uint32_t *__restrict__ link = prog + (ranges ? ranges + 2 * n_in : outs + 64)[0];
        __syncthreads();  // every wave has read its part of the digest out of `lds`, which the next message overwrites
        rec = prog + link[0];
        src = link + 1;
    }
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

// ------------------------------------------------------------------------------------------ witness-map digest
// SURVEY 8d (config 5): callers that do not want the full map back keep the return witnesses and a 32-byte digest per instance.
// Definition (include/acvm_amd.h acvm_batch_digest): D = sum over the ASSIGNED witnesses w of (value_w * g^(w+1) + h^(w+1)) in BN254-Fr, with
// two fixed field elements g and h; digest = Blake2s-256(D as 32 big-endian bytes). A polynomial fingerprint: one field product per
// witness, order-free, and linear -- so that (1) a witness the level kernels keep scaled needs no product of its own to leave the scaled
// form: its stored value scale_w * value_w is multiplied by g^(w+1) / scale_w, a circuit constant; (2) two products share one Montgomery
// reduction (fr29_dot<2>); (3) the h-sum of an instance the level kernels solved is a constant of the plan (its assigned set is the
// planner's), added once at the end. Round 2's digest (one Blake2s compression per pair of witnesses plus a product to leave the scaled
// form) cost 100 ms per 4 096 x 10^6 tile, two thirds of a solve; this one is bound by reading the table once.
// Partial sums: a lane covers DIGEST_CHUNK witnesses of one instance and stores their sum (Montgomery form, reduced) at partial[chunk][lane];
// digest_final_kernel adds the rows up. No atomics.
static constexpr uint32_t DIGEST_CHUNK = 1024;
struct FpAcc {
    Fr29 h;
    uint32_t hw;
};
__device__ __forceinline__ void fp_init(FpAcc &a) {
#pragma unroll
    for (int i = 0; i < 9; i++) a.h.v[i] = 0;
    a.hw = 0;
}
__device__ __forceinline__ void fp_add(FpAcc &a, const Fr29 &x, uint32_t weight) {
    gate_h_room(a.h, a.hw, weight);
    a.h = fr29_addl(a.h, x);
}
__device__ __forceinline__ Fr fp_value(const FpAcc &a) {
    GateSum s;
    s.v = fr29_norm(a.h);
    s.bound = a.hw;
    return fr29_pack(gate_sum_canon(s));
}
// sum of stored_w * coef_w over the listed witnesses for a wave whose lanes are all instances of the level kernels (wave-uniform
// coefficients, FP_DOT products per Montgomery reduction: 81 multiply-adds per product + 81 per reduction, so four to a reduction are 101 per
// term against the 121.5 of pairs and the 162 of single products; four rows in flight per lane). next: produces (witness row, coefficient address) pairs.
static constexpr int FP_DOT = 4;
template <class Next>
__device__ __forceinline__ Fr fp_sum_generic(const uint4 *__restrict__ W, uint64_t Bp, uint64_t j, Next next) {
    FpAcc acc;
    fp_init(acc);
    for (;;) {
        uint32_t row[FP_DOT];
        const uint32_t *c[FP_DOT];
        int n = 0;
        while (n < FP_DOT && next(row[n], c[n])) n++;  // wave-uniform
        if (n == 0) break;
        Fr29 x[FP_DOT], k[FP_DOT];
#pragma unroll
        for (int t = 0; t < FP_DOT; t++)
            if (t < n) {
                x[t] = fr29_from(fr_load_nt(W, row[t], Bp, j));
                k[t] = fr29_from(fr_const(c[t], 0));
            }
        if (n == FP_DOT) {
            fp_add(acc, fr29_dot<FP_DOT>(x, k), 17);
            continue;
        }
        if (n == 3) {  // (six to a reduction measured no faster than four, at 132 VGPRs)
            const Fr29 l[3] = {x[0], x[1], x[2]}, m[3] = {k[0], k[1], k[2]};
            fp_add(acc, fr29_dot<3>(l, m), 17);
        } else if (n == 2) {
            const Fr29 l[2] = {x[0], x[1]}, m[2] = {k[0], k[1]};
            fp_add(acc, fr29_dot<2>(l, m), 17);
        } else fp_add(acc, fr29_mul(x[0], k[0]), 17);
        break;
    }
    return fp_value(acc);
}
// after the solve, from the table: lane = instance first + t (or ids[t]), blockIdx.y = chunk of witnesses. Instances of the exact path (event
// word set; `assigned` bitmap, lane slow_index[j]) hold plain values and carry their own h-sum.
__global__ void __launch_bounds__(256) digest_chunk_kernel(const uint4 *__restrict__ W, uint64_t Bp, uint32_t first, uint32_t n, uint32_t n_witnesses,
                                                           const uint32_t *__restrict__ producer, const Unscale u, const DigestTables T,
                                                           const int32_t *__restrict__ slow_index, const uint32_t *__restrict__ assigned, uint32_t n_slow,
                                                           uint4 *__restrict__ partial, uint32_t chunk0) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, chunk = chunk0 + blockIdx.y;
    const bool live = t < n;
    const uint64_t j = (uint64_t)first + (live ? t : 0u);
    const bool generic = !u.event || u.event[j] == 0xFFFFFFFFu;  // solved by the level kernels: the planner's assigned set, scaled columns
    const uint32_t w_begin = chunk * DIGEST_CHUNK, w_end = min(n_witnesses, w_begin + DIGEST_CHUNK);
    Fr sum;
    if (__builtin_amdgcn_ballot_w64(live && !generic) == 0) {
        uint32_t w = w_begin;
        sum = fp_sum_generic(W, Bp, j, [&](uint32_t &row, const uint32_t *&coef) {
            while (w < w_end && producer[w] == 0xFFFFFFFFu) w++;
            if (w >= w_end) return false;
            const uint32_t ui = u.index ? u.index[w] : 0xFFFFFFFFu;
            row = w;
            coef = ui != 0xFFFFFFFFu ? T.g_scaled + 8 * (uint64_t)ui : T.g_pow + 8 * (uint64_t)w;
            w++;
            return true;
        });
    } else {  // a wave that holds an instance of the exact path: per-lane coefficients and masks
        const uint32_t lane = generic ? 0u : (uint32_t)slow_index[j];
        sum = fr_zero();
        for (uint32_t w = w_begin; w < w_end; w++) {
            const bool present = generic ? producer[w] != 0xFFFFFFFFu : ((assigned[(uint64_t)(w >> 5) * n_slow + lane] >> (w & 31)) & 1u) != 0u;
            if (!present) continue;
            const uint32_t ui = generic && u.index ? u.index[w] : 0xFFFFFFFFu;
            const Fr k = ui != 0xFFFFFFFFu ? fr_const(T.g_scaled, ui) : fr_const(T.g_pow, w);
            sum = fr_add(sum, fr_mul(fr_load(W, w, Bp, j), k));
            if (!generic) sum = fr_add(sum, fr_const(T.h_pow, w));
        }
    }
    if (live) fr_store(partial, chunk, n, t, sum);
}
// The same sums DURING the solve (PlanOpts::fold_digest): records [PK_DIGEST_LEAF, row of `partial`, n, (witness, row of g_scaled or NONE) x n] of the
// witnesses that are complete (the planner's assigned set = the generic instance), read through the row map of slot reuse. Instances
// that leave the generic path are summed from the exact path's map afterwards.
__global__ void __launch_bounds__(256) digest_fold_level_kernel(const uint4 *__restrict__ W, uint64_t Bp, uint32_t B, const uint32_t *__restrict__ prog,
                                                                const uint32_t *__restrict__ offsets, const uint32_t *__restrict__ slot_of,
                                                                const DigestTables T, uint4 *__restrict__ partial) {
    const uint64_t j0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = j0 < B;
    const uint64_t j = live ? j0 : 0u;
    const uint32_t *__restrict__ r = prog + offsets[blockIdx.y];
    const uint32_t n = r[2];
    uint32_t i = 0;
    const Fr sum = fp_sum_generic(W, Bp, j, [&](uint32_t &row, const uint32_t *&coef) {
        if (i >= n) return false;
        const uint32_t w = r[3 + 2 * i], ui = r[4 + 2 * i];
        row = slot_of ? slot_of[w] : w;
        coef = ui != 0xFFFFFFFFu ? T.g_scaled + 8 * (uint64_t)ui : T.g_pow + 8 * (uint64_t)w;
        i++;
        return true;
    });
    if (live) fr_store(partial, r[1], Bp, j, sum);
}
// digest = Blake2s-256(D): D = the rows of `partial` added up (+ the plan's h-sum for an instance of the level kernels), as 32 big-endian
// bytes. partial laid out [row][2 halves][stride]; instances [first, first + n) of it.
__global__ void __launch_bounds__(128) digest_final_kernel(const uint4 *__restrict__ partial, uint32_t n_rows, uint64_t stride, uint32_t first, uint32_t n,
                                                           const uint32_t *__restrict__ event, uint32_t event_first, const DigestTables T, uint8_t *__restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    Fr s = fr_zero();
    for (uint32_t r = 0; r < n_rows; r++) s = fr_add(s, fr_load(partial, r, stride, (uint64_t)first + t));
    const bool generic = !event || event[(uint64_t)event_first + t] == 0xFFFFFFFFu;
    if (generic) s = fr_add(s, fr_const(T.h_generic, 0));
    const Fr c = fr_to_canonical(s);
    uint32_t h[8], w[16];
    blake2s_init(h);
#pragma unroll
    for (int k = 0; k < 8; k++) { w[k] = bswap32(c.v[7 - k]); w[8 + k] = 0u; }  // big-endian bytes as little-endian message words
    blake2s_compress(h, w, 32u, true);
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int k = 0; k < 4; k++) out[(uint64_t)t * 32 + 4 * i + k] = (uint8_t)(h[i] >> (8 * k));
}
uint32_t digest_chunks(uint32_t n_witnesses) { return (n_witnesses + DIGEST_CHUNK - 1) / DIGEST_CHUNK; }
// partial: scratch of digest_chunks(n_witnesses) x n x 32 bytes
void launch_digest(hipStream_t s, const uint4 *W, uint64_t Bp, uint32_t first, uint32_t n, uint32_t n_witnesses, const uint32_t *producer, const Unscale &u,
                   const DigestTables &T, const int32_t *slow_index, const uint32_t *assigned, uint32_t n_slow, uint4 *partial, uint8_t *out) {
    if (!n) return;
    const uint32_t n_chunks = digest_chunks(n_witnesses);
    for (uint32_t done = 0; done < n_chunks; done += 65535u) {  // gridDim.y is limited to 65535
        const uint32_t m = n_chunks - done > 65535u ? 65535u : n_chunks - done;
        hipLaunchKernelGGL(digest_chunk_kernel, dim3((n + 255) / 256, m), dim3(256), 0, s, W, Bp, first, n, n_witnesses, producer, u, T, slow_index, assigned, n_slow,
                           partial, done);
    }
    hipLaunchKernelGGL(digest_final_kernel, dim3((n + 127) / 128), dim3(128), 0, s, partial, n_chunks, (uint64_t)n, 0u, n, u.event, first, T, out);
}
void launch_digest_fold_level(hipStream_t s, const uint4 *W, uint64_t Bp, uint32_t B, const DeviceProgram &dp, const uint32_t *offsets, uint32_t n,
                              const DigestTables &T, uint4 *partial) {
    if (!n || !B) return;
    for (uint32_t done = 0; done < n; done += 65535u) {
        const uint32_t m = n - done > 65535u ? 65535u : n - done;
        hipLaunchKernelGGL(digest_fold_level_kernel, dim3((B + 255) / 256, m), dim3(256), 0, s, W, Bp, B, dp.prog, offsets + done, dp.slot_of, T, partial);
    }
}
void launch_digest_final(hipStream_t s, const uint4 *partial, uint32_t n_rows, uint64_t stride, uint32_t first, uint32_t n, const uint32_t *event, const DigestTables &T,
                         uint8_t *out) {
    if (!n) return;
    hipLaunchKernelGGL(digest_final_kernel, dim3((n + 127) / 128), dim3(128), 0, s, partial, n_rows, stride, first, n, event, first, T, out);
}

}  // namespace acvm
