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
// level separately). A block of WAVES waves serves 64 instances of one record -- and of the records chained behind it (plan.cpp "hash
// chains": the hash of another hash's digest runs in its predecessor's block and takes the digest bytes from LDS).
//
// What bounds the launch (round 4, tools/t_hash_sweep.py): at 2^16 instances config 3 is 1 024 blocks for 1 024 SIMD slots, all resident at
// once, so the whole chip moves through read / hash / write in lock-step -- HBM idles while every block hashes, the ALUs idle while every
// block waits for its rows: 87 us = 59 us of traffic at the streaming rate + 19 us of dependent hash chain + the ramp. A launch that holds
// more items than slots overlaps by itself; this one has to overlap INSIDE the block:
//   * wave 0 hashes; while it does, waves 1..3 (a) store the outputs of the PREVIOUS record of the chain from its digest in LDS and (b)
//     fetch the inputs of the NEXT record that are not digest bytes (the planner admits a record to a chain only if those were known when
//     the head was launched) into the second message buffer. Config 3: the 32 extra rows of the Keccak travel under the SHA-256, the 32
//     output rows of the SHA-256 under the Keccak; only the head's 64 input rows and the tail's 32 output rows are moved by all four waves
//     with nothing to hide behind.
//   * the byte <-> Montgomery-form tables (8 KiB + 1 KiB) live in LDS: a fetched row is recognised as a byte by one table byte and a 32-byte
//     compare (28 instructions per row instead of the 80 of the arithmetic form, which round 2 had taken because a wave-wide gather from
//     the 8 KiB table in memory is served by the CU's one vector L1 at a line per cycle), an output byte is converted by a 32-byte LDS read.
//   * the hash bodies name gfx950's three-input bit operation and funnel shift (hash_device.hpp): SHA-256 24 instead of 29 instructions per
//     round, Keccak-f 180 instead of 288.
// the RANGE opcode fused on input i of the record fails: its opcode index, else 0xFFFFFFFF (is_byte: the value is the byte `low`)
__device__ __forceinline__ uint32_t range_check(const uint32_t *__restrict__ ranges, uint32_t i, bool is_byte, uint32_t low) {
    const uint32_t op = ranges[2u * i], bits = ranges[2u * i + 1u];
    if (op == 0xFFFFFFFFu) return op;
    return is_byte && (low >> bits) == 0u ? 0xFFFFFFFFu : op;
}
// the byte tables of ops_common.hpp in LDS: mont[2 d], mont[2 d + 1] = the stored form of byte d; key = BYTE_KEY as words
struct CoopTables {
    const uint4 *mont;
    const uint32_t *key;
};
__device__ __forceinline__ uint32_t coop_low_limb(const Fr &a, const CoopTables &T, bool &is_byte) {
    const uint32_t k = a.v[0] & 1023u;
    uint32_t d = (T.key[k >> 2] >> (8u * (k & 3u))) & 0xffu;  // the only byte this value can be (ops_common.hpp fr_is_byte)
    const uint4 lo = T.mont[2u * d], hi = T.mont[2u * d + 1u];
    const uint32_t diff = ((a.v[0] ^ lo.x) | (a.v[1] ^ lo.y) | (a.v[2] ^ lo.z) | (a.v[3] ^ lo.w)) | ((a.v[4] ^ hi.x) | (a.v[5] ^ hi.y) | (a.v[6] ^ hi.z) | (a.v[7] ^ hi.w));
    is_byte = diff == 0u;
    if (!is_byte) d = fr29_redc_low(fr29_from(a));
    return d;
}
__device__ __forceinline__ Fr coop_from_byte(uint32_t d, const CoopTables &T) {
    const uint4 lo = T.mont[2u * d], hi = T.mont[2u * d + 1u];
    return Fr{{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w}};
}
// An input of a record on its way into the message: the witness's row -- or, for an initial witness with a byte plane (plan.hpp "Byte planes"), the
// 4-byte word the import left beside the row: the low limb and the is-byte flag coop_low_limb would form from the row's 32 bytes.
struct BytePlanes {
    const uint32_t *of;     // witness -> plane or NONE (null: no planes)
    const uint32_t *plane;  // [plane][instance]
};
struct CoopIn {
    Fr row;
    uint32_t word;
    bool planar;  // wave-uniform
};
__device__ __forceinline__ CoopIn coop_get(const FastPolicy &p, uint32_t w, const BytePlanes &bp) {
    const uint32_t pl = bp.of ? bp.of[w] : 0xFFFFFFFFu;  // (a scalar load: w is wave-uniform)
    CoopIn a;
    a.planar = pl != 0xFFFFFFFFu;
    a.word = 0;
    if (a.planar) a.word = __builtin_nontemporal_load(bp.plane + (uint64_t)pl * p.Bp + p.j);
    else a.row = p.load(w);
    return a;
}
__device__ __forceinline__ uint32_t coop_take(const CoopIn &a, const CoopTables &T, bool &is_byte) {
    if (!a.planar) return coop_low_limb(a.row, T, is_byte);
    is_byte = (a.word >> 31) != 0u;
    return a.word & 0x1fffffffu;
}
// Rows of a record into its LDS message, four in flight per lane: the inputs list[k] (list == nullptr: input k itself) for k = w, w + nw, ... < count
// (w, nw, count, the list and the record are wave-uniform). Fused RANGE checks ride along.
__device__ __forceinline__ void coop_fetch_rows(const FastPolicy &p, const uint32_t *__restrict__ ins, const uint32_t *__restrict__ ranges,
                                                const uint32_t *__restrict__ list, uint32_t count, uint32_t w, uint32_t nw, uint8_t *bytes, uint32_t lane,
                                                const CoopTables &T, uint32_t &range_bad, const BytePlanes &bp) {
    for (uint32_t k = w; k < count; k += 4u * nw) {
        const uint32_t k1 = k + nw, k2 = k + 2u * nw, k3 = k + 3u * nw;
        const uint32_t i0 = list ? list[k] : k, i1 = k1 < count ? (list ? list[k1] : k1) : 0u, i2 = k2 < count ? (list ? list[k2] : k2) : 0u,
                       i3 = k3 < count ? (list ? list[k3] : k3) : 0u;
        CoopIn a0 = coop_get(p, ins[2u * i0], bp), a1 = a0, a2 = a0, a3 = a0;
        if (k1 < count) a1 = coop_get(p, ins[2u * i1], bp);
        if (k2 < count) a2 = coop_get(p, ins[2u * i2], bp);
        if (k3 < count) a3 = coop_get(p, ins[2u * i3], bp);
        bool b0, b1 = true, b2 = true, b3 = true;  // the value is a byte (then l is that byte)
        const uint32_t l0 = coop_take(a0, T, b0);
        bytes[4u * ((i0 >> 2) * 64u + lane) + (i0 & 3u)] = (uint8_t)l0;
        if (ranges) range_bad = min(range_bad, range_check(ranges, i0, b0, l0));
        if (k1 < count) {
            const uint32_t l1 = coop_take(a1, T, b1);
            bytes[4u * ((i1 >> 2) * 64u + lane) + (i1 & 3u)] = (uint8_t)l1;
            if (ranges) range_bad = min(range_bad, range_check(ranges, i1, b1, l1));
        }
        if (k2 < count) {
            const uint32_t l2 = coop_take(a2, T, b2);
            bytes[4u * ((i2 >> 2) * 64u + lane) + (i2 & 3u)] = (uint8_t)l2;
            if (ranges) range_bad = min(range_bad, range_check(ranges, i2, b2, l2));
        }
        if (k3 < count) {
            const uint32_t l3 = coop_take(a3, T, b3);
            bytes[4u * ((i3 >> 2) * 64u + lane) + (i3 & 3u)] = (uint8_t)l3;
            if (ranges) range_bad = min(range_bad, range_check(ranges, i3, b3, l3));
        }
    }
}
// the inputs of a chained record that are bytes of its predecessor's digest (src[i] != NONE), i = w, w + nw, ...: LDS to LDS
__device__ __forceinline__ void coop_copy_chained(const uint32_t *__restrict__ src, const uint32_t *__restrict__ ranges, uint32_t n_in, uint32_t w, uint32_t nw,
                                                  const uint32_t (*dig)[64], uint8_t *bytes, uint32_t lane, uint32_t &range_bad) {
    for (uint32_t i = w; i < n_in; i += nw) {
        const uint32_t from = src[i];
        if (from == 0xFFFFFFFFu) continue;
        const uint32_t byte = (dig[from >> 2][lane] >> (8u * (from & 3u))) & 0xffu;
        bytes[4u * ((i >> 2) * 64u + lane) + (i & 3u)] = (uint8_t)byte;
        if (ranges) range_bad = min(range_bad, range_check(ranges, i, true, byte));
    }
}
// outputs i = w, w + nw, ... < 32 of a record from its digest in LDS; false on a conflict with an assigned output (hash.rs:92-103 stops at
// the first one; the flagged instance re-runs exactly)
__device__ __forceinline__ bool coop_store_outputs(const FastPolicy &p, const uint32_t *__restrict__ outs, const uint32_t (*dig)[64], uint32_t w, uint32_t nw,
                                                   uint32_t lane, const CoopTables &T) {
    bool ok = true;
    for (uint32_t i = w; i < 32u; i += nw) {
        const uint32_t byte = (dig[i >> 2][lane] >> (8u * (i & 3u))) & 0xffu;
        ok = p.insert(outs[2u * i], coop_from_byte(byte, T), outs[2u * i + 1u]) && ok;
    }
    return ok;
}
// WAVES = 4: one item -- 64 instances of a record and of its chain -- per block, pipelined as above. WAVES = 1, for launches that fill the
// chip anyhow (many records per level): four independent items per block, one per wave, which share only the tables; a wave does every
// phase in turn and talks to no other wave (a lane reads back only the bytes and words it wrote itself).
// LDS of an item: n_buf message buffers of buf_words x 64 words (a second one where the next record's rows are fetched while this one is
// hashed; launches whose longest message leaves no room for it take turns), then the digest of the record being hashed and, pipelined, of
// its predecessor.
template <int WAVES>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8)))
hash_coop_level_kernel(uint4 *W, uint64_t Bp, uint32_t B, const uint32_t *__restrict__ offsets, uint32_t *__restrict__ event,
                       const uint32_t *__restrict__ prog, const uint32_t *__restrict__ slot_of, uint32_t buf_words, uint32_t n_buf,
                       const uint32_t *__restrict__ plane_of, const uint32_t *__restrict__ plane) {
    extern __shared__ uint32_t lds[];
    const BytePlanes bp{plane_of, plane};
    __shared__ uint4 lds_mont[512];
    __shared__ uint32_t lds_key[256];
    constexpr bool PIPE = WAVES > 1;
    // (the wave index as a scalar: everything indexed by it -- record words, witness ids, rows of slot_of -- is then a scalar load; as a
    // vector value each of those was a memory round trip of its own in front of every row)
    const uint32_t lane = threadIdx.x & 63u, q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t w = PIPE ? q : 0u;  // this wave's index among the WAVES waves of its item
    const uint32_t msg_words = n_buf * buf_words * 64u;
    uint32_t *item = lds + (PIPE ? 0u : q * (msg_words + 512u));
    uint32_t (*lds_dig)[8][64] = (uint32_t (*)[8][64])(item + msg_words);
    const uint64_t group = PIPE ? (uint64_t)blockIdx.x : (uint64_t)blockIdx.x * 4u + q;
    const uint64_t j = group * 64u + lane;
    for (uint32_t t = threadIdx.x; t < 512u; t += 256u) lds_mont[t] = ((const uint4 *)BYTE_MONT)[t];
    lds_key[threadIdx.x] = ((const uint32_t *)BYTE_KEY)[threadIdx.x];
    const CoopTables T{lds_mont, lds_key};
    __syncthreads();  // the tables (the only block-wide barrier of the one-wave items)
    if (!PIPE && group * 64u >= B) return;
    auto sync = [] { if (PIPE) __syncthreads(); };
    // (prog and slot_of arrive as kernel arguments of their own, not inside a struct: only a noalias argument lets the compiler read the record
    // with scalar loads; through the struct every record word was a vector load that also waited for the stores before it)
    const uint32_t *__restrict__ rec = prog + offsets[blockIdx.y];
    FastPolicy p{W, Bp, j, slot_of};
    const bool live = j < B;  // (rows are padded to Bp: the loads of a dead lane stay inside the table)
    const bool two = PIPE && n_buf == 2u;
    uint32_t cur = 0, dcur = 0;  // message buffer / digest buffer of `rec`
    uint32_t range_bad = 0xFFFFFFFFu, conflict = 0xFFFFFFFFu;
    const uint32_t *__restrict__ pend_outs = nullptr;  // outputs of the predecessor, still to be stored (pipelined items)
    uint32_t pend_opcode = 0;
    {   // the head's inputs: every wave of the item fetches
        const uint32_t n_in = rec[3];
        const uint32_t *ins = rec + 6, *ranges = (rec[2] & HASH_RANGE_FLAG) ? ins + 2u * n_in + 64u : nullptr;
        // (the bytes behind the message in its last word are never read: LdsMsg::word_le masks them)
        coop_fetch_rows(p, ins, ranges, nullptr, n_in, w, WAVES, (uint8_t *)item, lane, T, range_bad, bp);
    }
    sync();
    for (;;) {
        const uint32_t func = rec[2] & 0xffu, n_in = rec[3];
        const uint32_t *ins = rec + 6, *outs = ins + 2u * n_in;
        const uint32_t *ranges = (rec[2] & HASH_RANGE_FLAG) ? outs + 64u : nullptr;  // (opcode or NONE, bits) per input
        const bool more = (rec[2] & HASH_CHAIN_FLAG) != 0u;
        // the link to the record that hashes this digest: [its record, source of each of its inputs (byte of this digest or NONE), the count and the
        // indices of the NONE inputs]
        const uint32_t *__restrict__ link = more ? prog + (ranges ? ranges + 2u * n_in : outs + 64u)[0] : nullptr;
        const uint32_t *__restrict__ nrec = more ? prog + link[0] : nullptr;
        const uint32_t n_next = more ? nrec[3] : 0u;
        const uint32_t *nins = more ? nrec + 6 : nullptr, *nranges = more && (nrec[2] & HASH_RANGE_FLAG) ? nrec + 6 + 2u * n_next + 64u : nullptr;
        uint32_t *msg = item + cur * buf_words * 64u;
        uint32_t *nmsg = item + (two ? cur ^ 1u : 0u) * buf_words * 64u;
        if (w == 0) {
            Digest d;
            const LdsMsg m{msg, lane};
            if (func == 3u) d = sha256_body(m, n_in);
            else if (func == 4u) d = blake2s_body(m, n_in);
            else d = keccak256_body(m, n_in);
#pragma unroll
            for (int k = 0; k < 8; k++) lds_dig[dcur][k][lane] = d.d[k];
        } else if (PIPE) {  // beside the hash: the predecessor's outputs, the successor's rows
            if (pend_outs && live && !coop_store_outputs(p, pend_outs, lds_dig[dcur ^ 1u], w - 1u, WAVES - 1, lane, T)) conflict = min(conflict, pend_opcode);
            if (more && two) {
                coop_fetch_rows(p, nins, nranges, link + 1u + n_next + 1u, link[1u + n_next], w - 1u, WAVES - 1, (uint8_t *)nmsg, lane, T, range_bad, bp);
            }
        }
        sync();  // the digest is in LDS (and the message has been read)
        if (!more) break;
        if (!PIPE) {  // one wave: its outputs now
            if (live && !coop_store_outputs(p, outs, lds_dig[dcur], 0u, 1u, lane, T)) conflict = min(conflict, rec[1]);
        } else {
            pend_outs = outs;
            pend_opcode = rec[1];
        }
        // the successor's message: its rows (unless they came in beside the hash), then the digest bytes it reads
        if (!two) {
            coop_fetch_rows(p, nins, nranges, link + 1u + n_next + 1u, link[1u + n_next], w, WAVES, (uint8_t *)nmsg, lane, T, range_bad, bp);
        }
        coop_copy_chained(link + 1u, nranges, n_next, w, WAVES, lds_dig[dcur], (uint8_t *)nmsg, lane, range_bad);
        rec = nrec;
        cur = two ? cur ^ 1u : 0u;
        if (PIPE) dcur ^= 1u;
        sync();
    }
    // the tail's outputs (in a pipelined item the predecessor's went out beside the tail's hash): every wave of the item stores
    if (live) {
        const uint32_t *outs = rec + 6 + 2u * rec[3];
        if (!coop_store_outputs(p, outs, lds_dig[dcur], w, WAVES, lane, T)) conflict = min(conflict, rec[1]);
        const uint32_t bad = min(range_bad, conflict);
        if (bad != 0xFFFFFFFFu) flag_instance(event, j, bad);
    }
}
void launch_hash_coop_level(hipStream_t s, uint4 *W, uint64_t Bp, uint32_t B, const DeviceProgram &dp, const uint32_t *offsets, uint32_t n, uint32_t *event, uint32_t lds_words) {
    if (!n || !B) return;
    const uint32_t buf_words = std::max<uint32_t>(lds_words, 8u);
    const uint32_t groups_b = (B + 63u) / 64u;
    const uint64_t groups = (uint64_t)groups_b * n;  // items: one per 64 instances of a record
    // four waves per item while that still fits the chip about twice (1 024 SIMDs x 4-5 waves); a single item with a long message has the block's
    // LDS to itself either way
    const bool four = groups * 4u <= 8192u || buf_words > 32u;
    // pipelined items take a second message buffer while the block stays under 48 KiB of dynamic LDS (9 KiB are static: the tables)
    const uint32_t n_buf = four && (size_t)2 * buf_words * 256u <= (44u << 10) ? 2u : 1u;
    const size_t item_bytes = ((size_t)n_buf * buf_words * 64u + (four ? 1024u : 512u)) * 4u;
    for (uint32_t done = 0; done < n;) {  // gridDim.y is limited to 65535
        const uint32_t m = n - done > 65535u ? 65535u : n - done;
        if (four) hipLaunchKernelGGL(hash_coop_level_kernel<4>, dim3(groups_b, m), dim3(256), item_bytes, s, W, Bp, B, offsets + done, event, dp.prog, dp.slot_of, buf_words, n_buf, dp.byte_plane_of, dp.byte_plane);
        else hipLaunchKernelGGL(hash_coop_level_kernel<1>, dim3((groups_b + 3u) / 4u, m), dim3(256), 4u * item_bytes, s, W, Bp, B, offsets + done, event, dp.prog, dp.slot_of, buf_words, n_buf, dp.byte_plane_of, dp.byte_plane);
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
    Fr29 v = fr29_weak(fr29_norm(a.h));  // < 1.03 p; the last step down only when some lane's top limb says it may be needed
    if (__builtin_amdgcn_ballot_w64(v.v[8] >= fr_p29(8)) != 0) v = fr29_csub(v, 0);
    return fr29_pack(v);
}
// sum of stored_w * coef_w over the listed witnesses for a wave whose lanes are all instances of the level kernels (wave-uniform
// coefficients, FP_DOT products per Montgomery reduction: 81 multiply-adds per product + 81 per reduction, so four to a reduction are 101 per
// term against the 121.5 of pairs and the 162 of single products; four rows in flight per lane). next: produces (witness row, coefficient address) pairs.
#ifndef FP_DOT_N  // (tools/build_variant.sh -DFP_DOT_N=6: the A/B of NOTEBOOK.md section 9)
#define FP_DOT_N 4
#endif
static constexpr int FP_DOT = FP_DOT_N;
static_assert(FP_DOT >= 1 && FP_DOT <= 6, "fr29_dot's column accumulator budget");
template <int N>
__device__ __forceinline__ void fp_dot_first(FpAcc &acc, const Fr29 (&x)[FP_DOT], const Fr29 (&k)[FP_DOT]) {  // the first N of the loaded pairs
    Fr29 l[N], m[N];
#pragma unroll
    for (int t = 0; t < N; t++) { l[t] = x[t]; m[t] = k[t]; }
    fp_add(acc, fr29_dot<N>(l, m), 17);
}
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
        // the last, partial group (six to a reduction measured no faster than four in round 4, at 132 VGPRs, on the unfolded digest)
        if (FP_DOT > 5 && n == 5) fp_dot_first<(FP_DOT > 5 ? 5 : 1)>(acc, x, k);
        else if (FP_DOT > 4 && n == 4) fp_dot_first<(FP_DOT > 4 ? 4 : 1)>(acc, x, k);
        else if (FP_DOT > 3 && n == 3) fp_dot_first<(FP_DOT > 3 ? 3 : 1)>(acc, x, k);
        else if (FP_DOT > 2 && n == 2) fp_dot_first<(FP_DOT > 2 ? 2 : 1)>(acc, x, k);
        else fp_add(acc, fr29_mul(x[0], k[0]), 17);
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
// ---- the witness map hashed AS BYTES (include/acvm_amd.h acvm_batch_digest_blake2s: SURVEY 8d's "blake2s over the full witness vector", in tree
// form so that an instance's million witnesses are not one sequential chain): leaf k = Blake2s-256 over the 32-byte big-endian canonical values of
// witnesses [256 k, 256 k + 256), 0xFF x 32 for a witness the instance did not assign (no field element reads so); root = Blake2s-256 over
// u32_le(n_witnesses) and the leaves. One lane per (instance, leaf): 128 compressions and 256 canonicalising products; rows coalesced over the wave.
static constexpr uint32_t B2S_LEAF_WITNESSES = 256;
__device__ __forceinline__ void b2s_put_value(uint32_t (&w)[16], int half, bool present, const Fr &canon) {
#pragma unroll
    for (int k = 0; k < 8; k++) w[8 * half + k] = present ? bswap32(canon.v[7 - k]) : 0xFFFFFFFFu;  // big-endian bytes as little-endian message words
}
__global__ void __launch_bounds__(64) digest_b2s_leaf_kernel(const uint4 *__restrict__ W, uint64_t Bp, uint32_t first, uint32_t n, uint32_t n_witnesses,
                                                             const uint32_t *__restrict__ producer, const Unscale u, const int32_t *__restrict__ slow_index,
                                                             const uint32_t *__restrict__ assigned, uint32_t n_slow, uint32_t *__restrict__ leaves, uint32_t leaf0) {
    const uint32_t t = blockIdx.x * 64 + threadIdx.x, leaf = leaf0 + blockIdx.y;
    const bool live = t < n;
    const uint64_t j = (uint64_t)first + (live ? t : 0u);
    const bool generic = !u.event || u.event[j] == 0xFFFFFFFFu;  // solved by the level kernels: the planner's assigned set, scaled columns
    const uint32_t lane = generic ? 0u : (uint32_t)slow_index[j];
    const uint32_t w_begin = leaf * B2S_LEAF_WITNESSES, w_end = min(n_witnesses, w_begin + B2S_LEAF_WITNESSES);
    Fr one = fr_zero();
    one.v[0] = 1;
    uint32_t h[8], m[16];
    blake2s_init(h);
    const uint32_t n_pairs = (w_end - w_begin + 1) / 2;
    for (uint32_t q = 0; q < n_pairs; q++) {
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const uint32_t w = w_begin + 2 * q + half;
            bool present = false;
            Fr c = fr_zero();
            if (w < w_end) {
                present = generic ? producer[w] != 0xFFFFFFFFu : ((assigned[(uint64_t)(w >> 5) * n_slow + lane] >> (w & 31)) & 1u) != 0u;
                const uint32_t ui = generic && u.index ? u.index[w] : 0xFFFFFFFFu;
                // out of Montgomery form; a scaled column leaves it through the canonical integer 1 / scale instead of 1 (like export_witness_kernel)
                if (present) c = fr_mul(fr_load(W, w, Bp, j), ui != 0xFFFFFFFFu ? fr_const(u.consts_plain, ui) : one);
            }
            if (w < w_end) b2s_put_value(m, half, present, c);
            else {
#pragma unroll
                for (int k = 0; k < 8; k++) m[8 * half + k] = 0u;  // an odd last witness: 32 bytes of message in the block
            }
        }
        const bool last = q + 1 == n_pairs;
        const uint32_t bytes = 32u * (w_end - w_begin);
        blake2s_compress_body(h, m, last ? bytes : 64u * (q + 1), last);
    }
    if (live)
#pragma unroll
        for (int k = 0; k < 8; k++) leaves[((uint64_t)leaf * 8 + k) * n + t] = h[k];
}
__global__ void __launch_bounds__(64) digest_b2s_root_kernel(const uint32_t *__restrict__ leaves, uint32_t n_leaves, uint32_t n, uint32_t n_witnesses, uint8_t *__restrict__ out) {
    const uint32_t t = blockIdx.x * 64 + threadIdx.x;
    if (t >= n) return;
    // message: u32_le(n_witnesses), then the leaves (8 little-endian words each = their 32 digest bytes): 4 + 32 n_leaves bytes, streamed through a 16-word block
    uint32_t h[8], m[16];
    blake2s_init(h);
    const uint64_t total = 4ull + 32ull * n_leaves;
    m[0] = n_witnesses;
    uint32_t fill = 1;      // words of the block in use
    uint64_t done = 0;      // bytes compressed so far
    for (uint32_t leaf = 0; leaf < n_leaves; leaf++)
        for (int k = 0; k < 8; k++) {
            if (fill == 16) {  // (more data follows: not the last block)
                done += 64;
                blake2s_compress(h, m, (uint32_t)done, false);
                fill = 0;
            }
            // (fill is wave-uniform: a uniform-index store)
            const uint32_t v = leaves[((uint64_t)leaf * 8 + k) * n + t];
#pragma unroll
            for (int i = 0; i < 16; i++)
                if ((uint32_t)i == fill) m[i] = v;
            fill++;
        }
#pragma unroll
    for (int i = 0; i < 16; i++)
        if ((uint32_t)i >= fill) m[i] = 0u;
    blake2s_compress(h, m, (uint32_t)total, true);
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int k = 0; k < 4; k++) out[(uint64_t)t * 32 + 4 * i + k] = (uint8_t)(h[i] >> (8 * k));
}
uint32_t digest_b2s_leaves(uint32_t n_witnesses) { return (n_witnesses + B2S_LEAF_WITNESSES - 1) / B2S_LEAF_WITNESSES; }
void launch_digest_blake2s(hipStream_t s, const uint4 *W, uint64_t Bp, uint32_t first, uint32_t n, uint32_t n_witnesses, const uint32_t *producer, const Unscale &u,
                           const int32_t *slow_index, const uint32_t *assigned, uint32_t n_slow, uint32_t *leaves, uint8_t *out) {
    if (!n) return;
    const uint32_t n_leaves = digest_b2s_leaves(n_witnesses);
    for (uint32_t done = 0; done < n_leaves; done += 65535u) {  // gridDim.y is limited to 65535
        const uint32_t m = n_leaves - done > 65535u ? 65535u : n_leaves - done;
        hipLaunchKernelGGL(digest_b2s_leaf_kernel, dim3((n + 63) / 64, m), dim3(64), 0, s, W, Bp, first, n, n_witnesses, producer, u, slow_index, assigned, n_slow, leaves, done);
    }
    hipLaunchKernelGGL(digest_b2s_root_kernel, dim3((n + 63) / 64), dim3(64), 0, s, leaves, n_leaves, n, n_witnesses, out);
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
