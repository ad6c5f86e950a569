/*
 * oracle/hashes.c -- TEST INFRASTRUCTURE ONLY (CPU oracle). Not part of the shipped product path.
 *
 * SHA-256 (FIPS 180-4), Keccak-256 (original Keccak padding 0x01..0x80, rate 136) and
 * BLAKE2s-256 (RFC 7693, unkeyed), the three digests behind
 * /root/reference/blackbox_solver/src/lib.rs:47-60,86-99 (sha2 0.10.7 / sha3 0.10.8 `Keccak256` /
 * blake2 0.10.6 `Blake2s256`, all absent from the reference tree; public standards).
 * Pinned by tests/test_oracle_hashes.py: sha256("hello world") from
 * brillig_vm/src/black_box.rs:203-208, plus standard KATs and Python hashlib (externally pinned for
 * keccak/blake2s: the reference holds no fixed vector for them).
 */
#include "hashes.h"
#include <string.h>

/* ---------------- SHA-256 ---------------- */
static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
#define ROR32(x, n) (((x) >> (n)) | ((x) << (32 - (n))))

static void sha256_block(uint32_t h[8], const uint8_t *p) {
    uint32_t w[64];
    for (int i = 0; i < 16; i++)
        w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = ROR32(w[i - 15], 7) ^ ROR32(w[i - 15], 18) ^ (w[i - 15] >> 3);
        uint32_t s1 = ROR32(w[i - 2], 17) ^ ROR32(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; i++) {
        uint32_t S1 = ROR32(e, 6) ^ ROR32(e, 11) ^ ROR32(e, 25);
        uint32_t ch = (e & f) ^ (~e & g);
        uint32_t t1 = hh + S1 + ch + K256[i] + w[i];
        uint32_t S0 = ROR32(a, 2) ^ ROR32(a, 13) ^ ROR32(a, 22);
        uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
        uint32_t t2 = S0 + mj;
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

void oracle_sha256(const uint8_t *msg, size_t len, uint8_t out[32]) {
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    size_t full = len / 64;
    for (size_t i = 0; i < full; i++) sha256_block(h, msg + 64 * i);
    uint8_t tail[128];
    size_t rem = len - 64 * full;
    memset(tail, 0, sizeof tail);
    memcpy(tail, msg + 64 * full, rem);
    tail[rem] = 0x80;
    size_t tl = (rem + 9 <= 64) ? 64 : 128;
    uint64_t bits = (uint64_t)len * 8;
    for (int i = 0; i < 8; i++) tail[tl - 1 - i] = (uint8_t)(bits >> (8 * i));
    sha256_block(h, tail);
    if (tl == 128) sha256_block(h, tail + 64);
    for (int i = 0; i < 8; i++) {
        out[4 * i] = (uint8_t)(h[i] >> 24);
        out[4 * i + 1] = (uint8_t)(h[i] >> 16);
        out[4 * i + 2] = (uint8_t)(h[i] >> 8);
        out[4 * i + 3] = (uint8_t)h[i];
    }
}

/* ---------------- Keccak-256 ---------------- */
static const uint64_t KRC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
static const int KROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
#define ROL64(x, n) ((n) ? (((x) << (n)) | ((x) >> (64 - (n)))) : (x))

void oracle_keccak_f1600(uint64_t s[25]) {
    for (int r = 0; r < 24; r++) {
        uint64_t c[5], d[5], b[25];
        for (int x = 0; x < 5; x++) c[x] = s[x] ^ s[x + 5] ^ s[x + 10] ^ s[x + 15] ^ s[x + 20];
        for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ ROL64(c[(x + 1) % 5], 1);
        for (int i = 0; i < 25; i++) s[i] ^= d[i % 5];
        /* rho + pi: B[y, 2x+3y] = rot(A[x,y]) ; index = x + 5y */
        for (int x = 0; x < 5; x++)
            for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = ROL64(s[x + 5 * y], KROT[x + 5 * y]);
        for (int y = 0; y < 5; y++)
            for (int x = 0; x < 5; x++) s[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        s[0] ^= KRC[r];
    }
}

void oracle_keccak256(const uint8_t *msg, size_t len, uint8_t out[32]) {
    uint64_t s[25];
    memset(s, 0, sizeof s);
    const size_t rate = 136;
    size_t off = 0;
    while (len - off >= rate) {
        for (size_t i = 0; i < rate; i++) s[i / 8] ^= (uint64_t)msg[off + i] << (8 * (i % 8));
        oracle_keccak_f1600(s);
        off += rate;
    }
    uint8_t blk[136];
    memset(blk, 0, sizeof blk);
    memcpy(blk, msg + off, len - off);
    blk[len - off] ^= 0x01;
    blk[rate - 1] ^= 0x80;
    for (size_t i = 0; i < rate; i++) s[i / 8] ^= (uint64_t)blk[i] << (8 * (i % 8));
    oracle_keccak_f1600(s);
    for (int i = 0; i < 32; i++) out[i] = (uint8_t)(s[i / 8] >> (8 * (i % 8)));
}

/* ---------------- BLAKE2s-256 ---------------- */
static const uint32_t B2S_IV[8] = {0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19};
static const uint8_t B2S_SIGMA[10][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};

static void b2s_compress(uint32_t h[8], const uint8_t blk[64], uint64_t t, int last) {
    uint32_t m[16], v[16];
    for (int i = 0; i < 16; i++)
        m[i] = (uint32_t)blk[4 * i] | (uint32_t)blk[4 * i + 1] << 8 | (uint32_t)blk[4 * i + 2] << 16 | (uint32_t)blk[4 * i + 3] << 24;
    for (int i = 0; i < 8; i++) {
        v[i] = h[i];
        v[i + 8] = B2S_IV[i];
    }
    v[12] ^= (uint32_t)t;
    v[13] ^= (uint32_t)(t >> 32);
    if (last) v[14] = ~v[14];
#define G(a, b, c, d, x, y)          \
    v[a] = v[a] + v[b] + (x);        \
    v[d] = ROR32(v[d] ^ v[a], 16);   \
    v[c] = v[c] + v[d];              \
    v[b] = ROR32(v[b] ^ v[c], 12);   \
    v[a] = v[a] + v[b] + (y);        \
    v[d] = ROR32(v[d] ^ v[a], 8);    \
    v[c] = v[c] + v[d];              \
    v[b] = ROR32(v[b] ^ v[c], 7);
    for (int r = 0; r < 10; r++) {
        const uint8_t *s = B2S_SIGMA[r];
        G(0, 4, 8, 12, m[s[0]], m[s[1]]);
        G(1, 5, 9, 13, m[s[2]], m[s[3]]);
        G(2, 6, 10, 14, m[s[4]], m[s[5]]);
        G(3, 7, 11, 15, m[s[6]], m[s[7]]);
        G(0, 5, 10, 15, m[s[8]], m[s[9]]);
        G(1, 6, 11, 12, m[s[10]], m[s[11]]);
        G(2, 7, 8, 13, m[s[12]], m[s[13]]);
        G(3, 4, 9, 14, m[s[14]], m[s[15]]);
    }
#undef G
    for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
}

void oracle_blake2s(const uint8_t *msg, size_t len, uint8_t out[32]) {
    uint32_t h[8];
    for (int i = 0; i < 8; i++) h[i] = B2S_IV[i];
    h[0] ^= 0x01010020; /* digest 32, no key, fanout 1, depth 1 */
    size_t off = 0;
    while (len - off > 64) {
        b2s_compress(h, msg + off, (uint64_t)off + 64, 0);
        off += 64;
    }
    uint8_t blk[64];
    memset(blk, 0, 64);
    memcpy(blk, msg + off, len - off);
    b2s_compress(h, blk, (uint64_t)len, 1);
    for (int i = 0; i < 32; i++) out[i] = (uint8_t)(h[i / 4] >> (8 * (i % 4)));
}
