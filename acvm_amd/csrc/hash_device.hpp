// hash_device.hpp -- SHA-256 (FIPS 180-4), BLAKE2s-256 (RFC 7693) and Keccak-256 (Keccak-f[1600], rate 136, pad 0x01 .. 0x80) as the hash black
// boxes call them (blackbox_solver/src/lib.rs:47-65,86-99: sha2 0.10.7, blake2 0.10.6, sha3 0.10.8 Keccak256), over any message accessor
// M with `uint32_t word_le(uint32_t word_index, uint32_t len)` (little-endian word of the message, zero beyond len). Everything is
// __host__ __device__: tools/hash_device_host_test.hip runs these very routines on the host against hashlib and the oracle
// (tests/test_hash_device_on_host.py); on the device the three-input bit operations and the funnel shifts of gfx950 are named explicitly.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace acvm {

#define HASH_HD __host__ __device__ __forceinline__

struct Digest {
    uint32_t d[8];  // byte i of the digest at bits 8 * (i % 4) of d[i / 4]
    HASH_HD uint32_t byte(uint32_t i) const {
        uint32_t w = 0;
#pragma unroll
        for (int k = 0; k < 8; k++)
            if ((uint32_t)k == (i >> 2)) w = d[k];
        return (w >> (8u * (i & 3u))) & 0xffu;
    }
};

// ------------------------------------------------------------------------------------------------ SHA-256

HASH_HD uint32_t rotr32(uint32_t x, uint32_t n) { return __builtin_rotateright32(x, n); }
HASH_HD uint32_t bswap32(uint32_t x) { return __builtin_bswap32(x); }
// gfx950's three-input bit operation (v_bitop3_b32): bit i of the result = bit ((a_i << 2) | (b_i << 1) | c_i) of the truth table, i.e. the table
// of f is f(0xF0, 0xCC, 0xAA). The compiler finds it for some expressions only (it built SHA-256's sigmas from two v_xor each and Keccak's chi
// from v_bfi + v_xor), so the round functions name it: a ^ b ^ c, choose, majority, and Keccak's a ^ (~b & c).
#if defined(__HIP_DEVICE_COMPILE__)
HASH_HD uint32_t bit_xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
HASH_HD uint32_t bit_choose(uint32_t e, uint32_t f, uint32_t g) { return __builtin_amdgcn_bitop3_b32(e, f, g, 0xCA); }  // (e & f) | (~e & g)
HASH_HD uint32_t bit_majority(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8); }
HASH_HD uint32_t bit_chi(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xD2); }  // a ^ (~b & c)
// ({hi, lo} >> n) & 0xFFFFFFFF for 0 < n < 32 (v_alignbit_b32)
HASH_HD uint32_t funnel_shr(uint32_t hi, uint32_t lo, uint32_t n) { return __builtin_amdgcn_alignbit(hi, lo, n); }
#else  // the host-run tests
HASH_HD uint32_t bit_xor3(uint32_t a, uint32_t b, uint32_t c) { return a ^ b ^ c; }
HASH_HD uint32_t bit_choose(uint32_t e, uint32_t f, uint32_t g) { return (e & f) | (~e & g); }
HASH_HD uint32_t bit_majority(uint32_t a, uint32_t b, uint32_t c) { return (a & b) | (a & c) | (b & c); }
HASH_HD uint32_t bit_chi(uint32_t a, uint32_t b, uint32_t c) { return a ^ (~b & c); }
HASH_HD uint32_t funnel_shr(uint32_t hi, uint32_t lo, uint32_t n) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> n); }
#endif

// one 64-byte block: 24 instructions per round with the message schedule, 14 without (29 / 17 as the compiler built it from the plain formulas)
HASH_HD void sha256_compress(uint32_t (&h)[8], uint32_t (&w)[16]) {
    constexpr uint32_t SHA256_K_TABLE[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
        0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
        0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
        0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
        0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
        0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    uint32_t a = h[0], bb = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int r = 0; r < 64; r += 16) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (r) {
                const uint32_t w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
                const uint32_t s0 = bit_xor3(rotr32(w15, 7), rotr32(w15, 18), w15 >> 3);
                const uint32_t s1 = bit_xor3(rotr32(w2, 17), rotr32(w2, 19), w2 >> 10);
                w[i] = w[i] + s0 + w[(i + 9) & 15] + s1;
            }
            const uint32_t S1 = bit_xor3(rotr32(e, 6), rotr32(e, 11), rotr32(e, 25));
            const uint32_t t1 = hh + S1 + bit_choose(e, f, g) + SHA256_K_TABLE[r + i] + w[i];
            const uint32_t S0 = bit_xor3(rotr32(a, 2), rotr32(a, 13), rotr32(a, 22));
            const uint32_t t2 = S0 + bit_majority(a, bb, c);
            hh = g; g = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = t1 + t2;
        }
    }
    h[0] += a; h[1] += bb; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

template <class M>
HASH_HD Digest sha256_body(const M &m, uint32_t len) {
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
        sha256_compress(h, w);
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
HASH_HD void blake2s_compress_body(uint32_t (&h)[8], const uint32_t (&w)[16], uint32_t t, bool last) {
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
static inline __host__ __device__ __noinline__ void blake2s_compress(uint32_t (&h)[8], const uint32_t (&w)[16], uint32_t t, bool last) { blake2s_compress_body(h, w, t, last); }
HASH_HD void blake2s_init(uint32_t (&h)[8]) {
    const uint32_t IV[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
#pragma unroll
    for (int i = 0; i < 8; i++) h[i] = IV[i];
    h[0] ^= 0x01010020u;  // digest length 32, no key, fanout 1, depth 1
}
template <class M>
HASH_HD Digest blake2s_body(const M &m, uint32_t len) {
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
    HASH_HD void begin() { blake2s_init(h); n = 0; pend_full = false; have_lo = false; }
    HASH_HD void put(const uint32_t (&x)[8]) {
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
    HASH_HD void finish(uint32_t (&out)[8]) {
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

// Keccak-f[1600] on 32-bit halves (lo[i], hi[i] of lane i), in place: theta folds the column parities into the state with one three-input
// xor per half (s ^= C[x-1] ^ rotl(C[x+1], 1)), rho and pi walk the one 24-cycle of the lane permutation with v_alignbit_b32 (a 64-bit
// rotation is two of them), chi is one v_bitop3_b32 per half: 180 instructions per round where the compiler's translation of the 64-bit
// formulas took 288 (shifts and ors for the rotations, v_bfi + v_xor for chi). The round loop stays rolled (~75 live registers).
HASH_HD void rotl64_halves(uint32_t lo, uint32_t hi, int n, uint32_t &olo, uint32_t &ohi) {  // n: compile-time constant, 0 < n < 64
    if (n == 32) { olo = hi; ohi = lo; return; }
    if (n < 32) {
        olo = funnel_shr(lo, hi, 32 - n);
        ohi = funnel_shr(hi, lo, 32 - n);
    } else {
        olo = funnel_shr(hi, lo, 64 - n);
        ohi = funnel_shr(lo, hi, 64 - n);
    }
}
HASH_HD void keccak_f1600(uint32_t (&lo)[25], uint32_t (&hi)[25]) {
    constexpr uint64_t KECCAK_RC_TABLE[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL, 0x0000000080000001ULL,
        0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
        0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL,
        0x000000000000800aULL, 0x800000008000000aULL, 0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    constexpr int PILN[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
    constexpr int ROTC[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
#pragma unroll 1
    for (int round = 0; round < 24; round++) {
        {
            uint32_t Cl[5], Ch[5];
#pragma unroll
            for (int x = 0; x < 5; x++) {
                Cl[x] = bit_xor3(bit_xor3(lo[x], lo[x + 5], lo[x + 10]), lo[x + 15], lo[x + 20]);
                Ch[x] = bit_xor3(bit_xor3(hi[x], hi[x + 5], hi[x + 10]), hi[x + 15], hi[x + 20]);
            }
#pragma unroll
            for (int x = 0; x < 5; x++) {
                uint32_t rl, rh;
                rotl64_halves(Cl[(x + 1) % 5], Ch[(x + 1) % 5], 1, rl, rh);
                const uint32_t pl = Cl[(x + 4) % 5], ph = Ch[(x + 4) % 5];
#pragma unroll
                for (int y = 0; y < 25; y += 5) {
                    lo[x + y] = bit_xor3(lo[x + y], pl, rl);
                    hi[x + y] = bit_xor3(hi[x + y], ph, rh);
                }
            }
        }
        uint32_t tl = lo[1], th = hi[1];
#pragma unroll
        for (int i = 0; i < 24; i++) {
            const uint32_t bl = lo[PILN[i]], bh = hi[PILN[i]];
            rotl64_halves(tl, th, ROTC[i], lo[PILN[i]], hi[PILN[i]]);
            tl = bl;
            th = bh;
        }
#pragma unroll
        for (int y = 0; y < 25; y += 5) {
            const uint32_t a0 = lo[y], a1 = lo[y + 1], a2 = lo[y + 2], a3 = lo[y + 3], a4 = lo[y + 4];
            lo[y] = bit_chi(a0, a1, a2);
            lo[y + 1] = bit_chi(a1, a2, a3);
            lo[y + 2] = bit_chi(a2, a3, a4);
            lo[y + 3] = bit_chi(a3, a4, a0);
            lo[y + 4] = bit_chi(a4, a0, a1);
            const uint32_t b0 = hi[y], b1 = hi[y + 1], b2 = hi[y + 2], b3 = hi[y + 3], b4 = hi[y + 4];
            hi[y] = bit_chi(b0, b1, b2);
            hi[y + 1] = bit_chi(b1, b2, b3);
            hi[y + 2] = bit_chi(b2, b3, b4);
            hi[y + 3] = bit_chi(b3, b4, b0);
            hi[y + 4] = bit_chi(b4, b0, b1);
        }
        const uint64_t rc = KECCAK_RC_TABLE[round];
        lo[0] ^= (uint32_t)rc;
        hi[0] ^= (uint32_t)(rc >> 32);
    }
}

template <class M>
HASH_HD Digest keccak256_body(const M &m, uint32_t len) {
    uint32_t slo[25], shi[25];
#pragma unroll
    for (int i = 0; i < 25; i++) { slo[i] = 0; shi[i] = 0; }
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
            slo[i] ^= lo;
            shi[i] ^= hi;
        }
        keccak_f1600(slo, shi);
    }
    Digest out;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        out.d[2 * i] = slo[i];
        out.d[2 * i + 1] = shi[i];
    }
    return out;
}


}  // namespace acvm
