// ops_hash.hpp -- device routines of the hash black boxes, one lane per (instance, call):
//   SHA256 / Blake2s / Keccak256 / Keccak256VariableLength   acvm/src/pwg/blackbox/hash.rs:28-103
//   HashToField128Security                                    acvm/src/pwg/blackbox/hash.rs:13-24
//   the hash functions themselves                             blackbox_solver/src/lib.rs:47-65,86-99
//                                                             (sha2 0.10.7 / blake2 0.10.6 / sha3 0.10.8 Keccak256:
//                                                             FIPS 180-4, RFC 7693, Keccak-f[1600] rate 136 pad 0x01..0x80)
// The message is packed once into a per-lane byte buffer in device scratch (word-major [word][instance], so the 64
// lanes of a wave store and load 256 B contiguous per word) and hashed from there with a per-lane length, which
// covers the variable-length Keccak, the Brillig black-box ops and Schnorr's challenge with the same code.
#pragma once
#include "ops_common.hpp"

namespace acvm {

// per-lane byte buffer: word wi of instance j at base[wi * Bp + j]
struct MsgBuf {
    uint32_t *base;
    uint64_t Bp, j;
    uint32_t acc, pos;
    __device__ __forceinline__ void begin() { acc = 0; pos = 0; }
    __device__ __forceinline__ void put(uint32_t byte) {
        acc |= (byte & 0xffu) << (8u * (pos & 3u));
        pos++;
        if ((pos & 3u) == 0) { base[(uint64_t)((pos >> 2) - 1) * Bp + j] = acc; acc = 0; }
    }
    __device__ __forceinline__ void end() {
        if (pos & 3u) base[(uint64_t)(pos >> 2) * Bp + j] = acc;
    }
    __device__ __forceinline__ uint32_t word(uint32_t wi) const { return base[(uint64_t)wi * Bp + j]; }
    // little-endian word wi of the message zero-extended beyond len
    __device__ __forceinline__ uint32_t word_le(uint32_t wi, uint32_t len) const {
        if (4u * wi >= len) return 0u;
        uint32_t v = word(wi);
        const uint32_t k = len - 4u * wi;  // valid bytes
        if (k < 4u) v &= (1u << (8u * k)) - 1u;
        return v;
    }
};

// the same message held in LDS by a block that serves 64 instances (kernels_hash.hip hash_coop_level_kernel): word wi of lane l at
// words[wi * 64 + l], byte k of it at byte address 4 * (wi * 64 + l) + k -- conflict-free for both the byte writes and the word reads
struct LdsMsg {
    const uint32_t *words;
    uint32_t lane;
    __device__ __forceinline__ uint32_t word(uint32_t wi) const { return words[wi * 64u + lane]; }
    __device__ __forceinline__ uint32_t word_le(uint32_t wi, uint32_t len) const {
        if (4u * wi >= len) return 0u;
        uint32_t v = word(wi);
        const uint32_t k = len - 4u * wi;  // valid bytes
        if (k < 4u) v &= (1u << (8u * k)) - 1u;
        return v;
    }
};

struct Digest {
    uint32_t d[8];  // byte i of the digest at bits 8 * (i % 4) of d[i / 4]
    __device__ __forceinline__ uint32_t byte(uint32_t i) const {
        uint32_t w = 0;
#pragma unroll
        for (int k = 0; k < 8; k++)
            if ((uint32_t)k == (i >> 2)) w = d[k];
        return (w >> (8u * (i & 3u))) & 0xffu;
    }
};

// ------------------------------------------------------------------------------------------------ SHA-256
static __constant__ uint32_t SHA256_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

__device__ __forceinline__ uint32_t rotr32(uint32_t x, uint32_t n) { return __builtin_rotateright32(x, n); }
__device__ __forceinline__ uint32_t bswap32(uint32_t x) { return __builtin_bswap32(x); }

template <class M>
__device__ __forceinline__ Digest sha256_body(const M &m, uint32_t len) {
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    const uint32_t n_blocks = (len + 9u + 63u) / 64u;
    for (uint32_t b = 0; b < n_blocks; b++) {
        uint32_t w[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const uint32_t wi = 16u * b + i;
            uint32_t v = bswap32(m.word_le(wi, len));
            if (4u * wi <= len && len < 4u * wi + 4u) v |= 0x80u << (24u - 8u * (len - 4u * wi));  // the 1 bit after the message
            w[i] = v;
        }
        if (b == n_blocks - 1) {
            w[14] = len >> 29;  // bit length, big-endian 64-bit
            w[15] = len << 3;
        }
        uint32_t a = h[0], bb = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int r = 0; r < 64; r += 16) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (r) {
                    const uint32_t w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
                    const uint32_t s0 = rotr32(w15, 7) ^ rotr32(w15, 18) ^ (w15 >> 3);
                    const uint32_t s1 = rotr32(w2, 17) ^ rotr32(w2, 19) ^ (w2 >> 10);
                    w[i] = w[i] + s0 + w[(i + 9) & 15] + s1;
                }
                const uint32_t S1 = rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25);
                const uint32_t ch = (e & f) ^ (~e & g);
                const uint32_t t1 = hh + S1 + ch + SHA256_K[r + i] + w[i];
                const uint32_t S0 = rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22);
                const uint32_t mj = (a & bb) ^ (a & c) ^ (bb & c);
                const uint32_t t2 = S0 + mj;
                hh = g; g = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = t1 + t2;
            }
        }
        h[0] += a; h[1] += bb; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    Digest out;
#pragma unroll
    for (int i = 0; i < 8; i++) out.d[i] = bswap32(h[i]);  // digest bytes are the big-endian words
    return out;
}

// ------------------------------------------------------------------------------------------------ Blake2s-256 (RFC 7693)
#define B2S_G(a, b, c, d, x, y)          \
    a = a + b + (x); d = rotr32(d ^ a, 16); \
    c = c + d; b = rotr32(b ^ c, 12);       \
    a = a + b + (y); d = rotr32(d ^ a, 8);  \
    c = c + d; b = rotr32(b ^ c, 7);

// one compression: h <- F(h, 16 little-endian message words, byte counter t, final-block flag)
// (body; the shared out-of-line copy is blake2s_compress below, the digest kernels inline it: their h and w then stay in registers instead
// of travelling through the stack of a call)
__device__ __forceinline__ void blake2s_compress_body(uint32_t (&h)[8], const uint32_t (&w)[16], uint32_t t, bool last) {
    const uint32_t IV[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    uint32_t v[16];
#pragma unroll
    for (int i = 0; i < 8; i++) { v[i] = h[i]; v[8 + i] = IV[i]; }
    v[12] ^= t;
    if (last) v[14] = ~v[14];
    // fully unrolled: the message schedule indices are compile-time constants, w[] stays in registers
    constexpr uint8_t SIGMA[10][16] = {
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
        {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
        {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
        {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
        {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint32_t x[16];
#pragma unroll
        for (int i = 0; i < 16; i++) x[i] = w[SIGMA[r][i]];
        B2S_G(v[0], v[4], v[8], v[12], x[0], x[1]);
        B2S_G(v[1], v[5], v[9], v[13], x[2], x[3]);
        B2S_G(v[2], v[6], v[10], v[14], x[4], x[5]);
        B2S_G(v[3], v[7], v[11], v[15], x[6], x[7]);
        B2S_G(v[0], v[5], v[10], v[15], x[8], x[9]);
        B2S_G(v[1], v[6], v[11], v[12], x[10], x[11]);
        B2S_G(v[2], v[7], v[8], v[13], x[12], x[13]);
        B2S_G(v[3], v[4], v[9], v[14], x[14], x[15]);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[8 + i];
}
static inline __device__ __noinline__ void blake2s_compress(uint32_t (&h)[8], const uint32_t (&w)[16], uint32_t t, bool last) { blake2s_compress_body(h, w, t, last); }
__device__ __forceinline__ void blake2s_init(uint32_t (&h)[8]) {
    const uint32_t IV[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
#pragma unroll
    for (int i = 0; i < 8; i++) h[i] = IV[i];
    h[0] ^= 0x01010020u;  // digest length 32, no key, fanout 1, depth 1
}
template <class M>
__device__ __forceinline__ Digest blake2s_body(const M &m, uint32_t len) {
    uint32_t h[8];
    blake2s_init(h);
    const uint32_t n_blocks = len == 0 ? 1u : (len + 63u) / 64u;
    for (uint32_t b = 0; b < n_blocks; b++) {
        uint32_t w[16];
#pragma unroll
        for (int i = 0; i < 16; i++) w[i] = m.word_le(16u * b + i, len);
        const bool last = b == n_blocks - 1;
        blake2s_compress(h, w, last ? len : 64u * (b + 1), last);
    }
    Digest out;
#pragma unroll
    for (int i = 0; i < 8; i++) out.d[i] = h[i];  // little-endian words
    return out;
}
// Blake2s-256 over a stream of 32-byte pieces whose number is only known at the end (the witness-map digest of
// kernels_hash.hip): a full block is held back until more data arrives, because the last block is compressed differently
struct Blake2sPieces {
    uint32_t h[8], pend[16], lo[8];
    uint32_t n;  // pieces absorbed
    bool pend_full, have_lo;
    __device__ __forceinline__ void begin() { blake2s_init(h); n = 0; pend_full = false; have_lo = false; }
    __device__ __forceinline__ void put(const uint32_t (&x)[8]) {
        if (!have_lo) {
            if (pend_full) { blake2s_compress(h, pend, 32u * n, false); pend_full = false; }
#pragma unroll
            for (int i = 0; i < 8; i++) lo[i] = x[i];
            have_lo = true;
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) { pend[i] = lo[i]; pend[8 + i] = x[i]; }
            pend_full = true;
            have_lo = false;
        }
        n++;
    }
    __device__ __forceinline__ void finish(uint32_t (&out)[8]) {
        if (have_lo) {
            if (pend_full) blake2s_compress(h, pend, 32u * (n - 1), false);
#pragma unroll
            for (int i = 0; i < 8; i++) { pend[i] = lo[i]; pend[8 + i] = 0u; }
            blake2s_compress(h, pend, 32u * n, true);
        } else if (pend_full) {
            blake2s_compress(h, pend, 32u * n, true);
        } else {
#pragma unroll
            for (int i = 0; i < 16; i++) pend[i] = 0u;
            blake2s_compress(h, pend, 0u, true);
        }
#pragma unroll
        for (int i = 0; i < 8; i++) out[i] = h[i];
    }
};

// ------------------------------------------------------------------------------------------------ Keccak-256
static __constant__ uint64_t KECCAK_RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL, 0x0000000080000001ULL,
    0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL,
    0x000000000000800aULL, 0x800000008000000aULL, 0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};

__device__ __forceinline__ uint64_t rotl64(uint64_t x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }

// Keccak-f[1600] in place: theta, then rho and pi along the one 24-cycle of the lane permutation (two temporaries), then chi row by row
// (five temporaries): ~75 live registers instead of a second copy of the state (the first version needed 184 VGPRs and capped the class's
// kernels at two waves per SIMD). The round loop stays rolled.
__device__ __forceinline__ void keccak_f1600(uint64_t (&s)[25]) {
    constexpr int PILN[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
    constexpr int ROTC[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
#pragma unroll 1
    for (int round = 0; round < 24; round++) {
        {
            uint64_t C[5];
#pragma unroll
            for (int x = 0; x < 5; x++) C[x] = s[x] ^ s[x + 5] ^ s[x + 10] ^ s[x + 15] ^ s[x + 20];
#pragma unroll
            for (int x = 0; x < 5; x++) {
                const uint64_t D = C[(x + 4) % 5] ^ rotl64(C[(x + 1) % 5], 1);
#pragma unroll
                for (int y = 0; y < 25; y += 5) s[x + y] ^= D;
            }
        }
        uint64_t t = s[1];
#pragma unroll
        for (int i = 0; i < 24; i++) {
            const uint64_t b = s[PILN[i]];
            s[PILN[i]] = rotl64(t, ROTC[i]);
            t = b;
        }
#pragma unroll
        for (int y = 0; y < 25; y += 5) {
            const uint64_t a0 = s[y], a1 = s[y + 1], a2 = s[y + 2], a3 = s[y + 3], a4 = s[y + 4];
            s[y] = a0 ^ (~a1 & a2);
            s[y + 1] = a1 ^ (~a2 & a3);
            s[y + 2] = a2 ^ (~a3 & a4);
            s[y + 3] = a3 ^ (~a4 & a0);
            s[y + 4] = a4 ^ (~a0 & a1);
        }
        s[0] ^= KECCAK_RC[round];
    }
}

template <class M>
__device__ __forceinline__ Digest keccak256_body(const M &m, uint32_t len) {
    uint64_t s[25];
#pragma unroll
    for (int i = 0; i < 25; i++) s[i] = 0;
    const uint32_t n_blocks = len / 136u + 1u;  // the padding always adds at least one byte
    for (uint32_t b = 0; b < n_blocks; b++) {
#pragma unroll
        for (int i = 0; i < 17; i++) {
            const uint32_t wi = 34u * b + 2u * i;
            uint32_t lo = m.word_le(wi, len), hi = m.word_le(wi + 1, len);
            // pad10*1 with the Keccak (not SHA-3) domain byte: 0x01 right after the message, 0x80 on the last byte of the block
            const uint32_t p0 = 4u * wi, p1 = p0 + 4u;
            if (p0 <= len && len < p0 + 4u) lo |= 0x01u << (8u * (len - p0));
            if (p1 <= len && len < p1 + 4u) hi |= 0x01u << (8u * (len - p1));
            if (b == n_blocks - 1 && i == 16) hi |= 0x80000000u;
            s[i] ^= (uint64_t)hi << 32 | lo;
        }
        keccak_f1600(s);
    }
    Digest out;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        out.d[2 * i] = (uint32_t)s[i];
        out.d[2 * i + 1] = (uint32_t)(s[i] >> 32);
    }
    return out;
}

// one copy per kernel for the lane-per-instance callers (several call sites each in the exact kernels and the Brillig VM)
static inline __device__ __noinline__ Digest sha256_msg(const MsgBuf &m, uint32_t len) { return sha256_body(m, len); }
static inline __device__ __noinline__ Digest blake2s_msg(const MsgBuf &m, uint32_t len) { return blake2s_body(m, len); }
static inline __device__ __noinline__ Digest keccak256_msg(const MsgBuf &m, uint32_t len) { return keccak256_body(m, len); }

// digest bytes as a big-endian integer reduced mod p (from_be_bytes_reduce, generic_ark.rs:281-283), Montgomery form
__device__ __forceinline__ Fr digest_to_field(const Digest &d) {
    Fr c;
#pragma unroll
    for (int i = 0; i < 8; i++) c.v[i] = bswap32(d.d[7 - i]);  // limb i (LE) = big-endian bytes [28 - 4i, 32 - 4i)
    return fr_from_canonical(canon_reduce(c));
}

// ------------------------------------------------------------------------------------------------ the opcode
// [K_HASH, opcode, func, n_in, n_out, var_w, (w, num_bits) x n_in, (out, flag) x n_out]; func = BlackBoxFuncCall tag
// (3 SHA256, 4 Blake2s, 7 HashToField128Security, 11 Keccak256, 12 Keccak256VariableLength), | HASH_COOP_FLAG when every input is one
// byte wide, the function is 3, 4 or 11, n_out == 32 and 1 <= n_in <= 1024 (plan.cpp)
static constexpr uint32_t HASH_COOP_FLAG = 0x100u;
static constexpr uint32_t HASH_CHAIN_FLAG = 0x400u;  // level-schedule copy of the record: one more word behind it, the offset of the link to the record that hashes this digest (kernels_hash.hip)
static constexpr uint32_t HASH_RANGE_FLAG = 0x200u;  // level-schedule copy of the record: (RANGE opcode or NONE, bits) x n_in behind the outputs
template <class P>
__device__ __forceinline__ OpResult op_hash(const P &p, const uint32_t *__restrict__ r, uint32_t *scratch) {
    const uint32_t func = r[2] & 0xffu, n_in = r[3], n_out = r[4], var_w = r[5];  // bit 8: HASH_COOP_FLAG, a hint for the level kernel
    const uint32_t *ins = r + 6, *outs = ins + 2 * n_in;
    if (P::exact) {  // blackbox/mod.rs:55-62, get_inputs_vec order: inputs, then var_message_size
        for (uint32_t i = 0; i < n_in; i++)
            if (!p.known(ins[2 * i])) return op_fail(DE_MISSING_ASSIGNMENT, ins[2 * i]);
        if (var_w != K_NONE && !p.known(var_w)) return op_fail(DE_MISSING_ASSIGNMENT, var_w);
    }
    // get_hash_input (hash.rs:51-86): fetch_nearest_bytes = low ceil(num_bits / 8) bytes, least significant first
    MsgBuf m{scratch, p.Bp, p.j, 0u, 0u};
    m.begin();
    for (uint32_t i = 0; i < n_in;) {
        const uint32_t nb = (ins[2 * i + 1] + 7u) / 8u;
        if (nb > 32u) return op_fail_msg(DE_PANIC, 0, DM_FETCH_BYTES);  // slice end out of range (generic_ark.rs:316)
        // byte arrays, four witnesses at a time: four rows in flight and four independent reductions for the one wave a SIMD
        // holds at these batch sizes; only the low limb of each canonical value is formed (fr29_redc_low)
        if (nb == 1u && i + 4u <= n_in && (ins[2 * i + 3] + 7u) / 8u == 1u && (ins[2 * i + 5] + 7u) / 8u == 1u && (ins[2 * i + 7] + 7u) / 8u == 1u) {
            const Fr a0 = p.load(ins[2 * i]), a1 = p.load(ins[2 * i + 2]), a2 = p.load(ins[2 * i + 4]), a3 = p.load(ins[2 * i + 6]);
            bool b0, b1, b2, b3;
            const uint32_t l0 = fr_low_limb(a0, b0), l1 = fr_low_limb(a1, b1), l2 = fr_low_limb(a2, b2), l3 = fr_low_limb(a3, b3);
            m.put(l0);
            m.put(l1);
            m.put(l2);
            m.put(l3);
            i += 4u;
            continue;
        }
        const Fr cur = p.load(ins[2 * i]);
        i++;
        if (nb <= 3u) {
            const uint32_t low = fr29_redc_low(fr29_from(cur));
            for (uint32_t k = 0; k < nb; k++) m.put(low >> (8u * k));
            continue;
        }
        const Fr c = fr_to_canonical(cur);
#pragma unroll
        for (int k = 0; k < 32; k++)
            if ((uint32_t)k < nb) m.put(c.v[k >> 2] >> (8 * (k & 3)));
    }
    m.end();
    uint32_t len = m.pos;
    if (var_w != K_NONE) {  // hash.rs:68-83: `to_u128() as usize`
        const Fr c = fr_to_canonical(p.load(var_w));
        if (c.v[1] != 0u || c.v[0] > len) return op_fail_msg(DE_BLACKBOX_FAILED, 11u, DM_KECCAK_VAR_LEN, c.v[0], c.v[1]);
        len = c.v[0];
    }
    Digest d;
    if (func == 3u) d = sha256_msg(m, len);
    else if (func == 4u || func == 7u) d = blake2s_msg(m, len);
    else d = keccak256_msg(m, len);
    if (func == 7u) {  // solve_hash_to_field (hash.rs:13-24)
        if (!p.insert(outs[0], digest_to_field(d), outs[1])) return op_fail(DE_UNSATISFIED);
        return op_ok();
    }
    if (n_out != 32u) return op_fail_msg(DE_BLACKBOX_FAILED, func == 12u ? 11u : func, DM_HASH_OUTPUTS, n_out);  // hash.rs:92-97
    for (uint32_t i = 0; i < 32u; i++)
        if (!p.insert(outs[2 * i], fr_from_byte(d.byte(i)), outs[2 * i + 1])) return op_fail(DE_UNSATISFIED);
    return op_ok();
}

}  // namespace acvm
