/*
 * oracle/fr.c -- TEST INFRASTRUCTURE ONLY (CPU oracle). Not part of the shipped product path.
 * BN254-Fr arithmetic restating acir_field::FieldElement
 * (/root/reference/acir_field/src/generic_ark.rs). See fr.h for provenance and pinning.
 */
#include "fr.h"
#include <string.h>

typedef unsigned __int128 u128;

/* p = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001 */
const uint64_t FR_MODULUS[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL,
                                0x30644e72e131a029ULL};
static uint64_t N0INV;      /* -p^-1 mod 2^64 */
static fr_t R1, R2, R3;     /* R, R^2, R^3 mod p as raw limbs */
static fr_t C256;           /* 256 in Montgomery form */

static int geq_p(const uint64_t a[4]) {
    for (int i = 3; i >= 0; i--) {
        if (a[i] > FR_MODULUS[i]) return 1;
        if (a[i] < FR_MODULUS[i]) return 0;
    }
    return 1;
}
static uint64_t add4(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
        c += (u128)a[i] + b[i];
        r[i] = (uint64_t)c;
        c >>= 64;
    }
    return (uint64_t)c;
}
static uint64_t sub4(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
    uint64_t borrow = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a[i] - b[i] - borrow;
        r[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
    return borrow;
}
static void dbl_mod(uint64_t a[4]) {
    uint64_t c = add4(a, a, a);
    if (c || geq_p(a)) sub4(a, a, FR_MODULUS);
}

void fr_mul(fr_t *r, const fr_t *a, const fr_t *b) {
    /* CIOS Montgomery multiplication, 4x64 limbs */
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)a->l[j] * b->l[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (uint64_t)c;
        t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * N0INV;
        c = (u128)m * FR_MODULUS[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 4; j++) {
            c += (u128)m * FR_MODULUS[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (uint64_t)c;
        t[4] = t[5] + (uint64_t)(c >> 64);
    }
    if (t[4] || geq_p(t)) sub4(t, t, FR_MODULUS);
    memcpy(r->l, t, 32);
}

__attribute__((constructor)) static void fr_init(void) {
    /* Newton iteration for p^-1 mod 2^64 */
    uint64_t inv = 1;
    for (int i = 0; i < 6; i++) inv *= 2 - FR_MODULUS[0] * inv;
    N0INV = (uint64_t)0 - inv;
    uint64_t x[4] = {1, 0, 0, 0};
    for (int i = 0; i < 256; i++) dbl_mod(x);
    memcpy(R1.l, x, 32);
    for (int i = 0; i < 256; i++) dbl_mod(x);
    memcpy(R2.l, x, 32);
    fr_mul(&R3, &R2, &R2); /* R2*R2/R = R^3 */
    fr_from_u64(&C256, 256);
}

void fr_zero(fr_t *r) { memset(r, 0, sizeof *r); }
void fr_one(fr_t *r) { *r = R1; }
void fr_from_canonical(fr_t *r, const uint64_t in[4]) {
    fr_t t;
    memcpy(t.l, in, 32);
    fr_mul(r, &t, &R2);
}
void fr_from_u64(fr_t *r, uint64_t v) {
    uint64_t t[4] = {v, 0, 0, 0};
    fr_from_canonical(r, t);
}
void fr_to_canonical(const fr_t *a, uint64_t out[4]) {
    fr_t one = {{1, 0, 0, 0}}, t;
    fr_mul(&t, a, &one);
    memcpy(out, t.l, 32);
}

void fr_from_be_bytes_reduce(fr_t *r, const uint8_t *bytes, size_t len) {
    /* generic_ark.rs:281-283 -> ark-ff from_be_bytes_mod_order: the big-endian integer mod p */
    if (len <= 32) {
        uint64_t v[4] = {0, 0, 0, 0};
        for (size_t i = 0; i < len; i++) {
            size_t pos = len - 1 - i; /* byte significance */
            v[pos / 8] |= (uint64_t)bytes[i] << (8 * (pos % 8));
        }
        while (geq_p(v)) sub4(v, v, FR_MODULUS);
        fr_from_canonical(r, v);
        return;
    }
    fr_t acc, t, b;
    fr_zero(&acc);
    for (size_t i = 0; i < len; i++) {
        fr_mul(&t, &acc, &C256);
        fr_from_u64(&b, bytes[i]);
        fr_add(&acc, &t, &b);
    }
    *r = acc;
}

void fr_to_be_bytes(const fr_t *a, uint8_t out[32]) {
    uint64_t c[4];
    fr_to_canonical(a, c);
    for (int i = 0; i < 32; i++) out[31 - i] = (uint8_t)(c[i / 8] >> (8 * (i % 8)));
}

void fr_add(fr_t *r, const fr_t *a, const fr_t *b) {
    uint64_t t[4];
    uint64_t c = add4(t, a->l, b->l);
    if (c || geq_p(t)) sub4(t, t, FR_MODULUS);
    memcpy(r->l, t, 32);
}
void fr_sub(fr_t *r, const fr_t *a, const fr_t *b) {
    uint64_t t[4];
    if (sub4(t, a->l, b->l)) add4(t, t, FR_MODULUS);
    memcpy(r->l, t, 32);
}
int fr_is_zero(const fr_t *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
void fr_neg(fr_t *r, const fr_t *a) {
    if (fr_is_zero(a)) {
        fr_zero(r);
        return;
    }
    uint64_t t[4];
    sub4(t, FR_MODULUS, a->l);
    memcpy(r->l, t, 32);
}
int fr_is_one(const fr_t *a) { return memcmp(a->l, R1.l, 32) == 0; }
int fr_eq(const fr_t *a, const fr_t *b) { return memcmp(a->l, b->l, 32) == 0; }
int fr_cmp(const fr_t *a, const fr_t *b) {
    uint64_t x[4], y[4];
    fr_to_canonical(a, x);
    fr_to_canonical(b, y);
    for (int i = 3; i >= 0; i--) {
        if (x[i] < y[i]) return -1;
        if (x[i] > y[i]) return 1;
    }
    return 0;
}

/* ---- inversion: binary extended Euclid on the canonical integers (ark-ff uses the same family,
 * Guajardo-Kumar-Paar-Pelzl alg. 16); the result is the unique inverse so any algorithm matches. */
static int is_even(const uint64_t a[4]) { return (a[0] & 1) == 0; }
static void shr1(uint64_t a[4], uint64_t carry_in) {
    a[0] = (a[0] >> 1) | (a[1] << 63);
    a[1] = (a[1] >> 1) | (a[2] << 63);
    a[2] = (a[2] >> 1) | (a[3] << 63);
    a[3] = (a[3] >> 1) | (carry_in << 63);
}
static int is_one4(const uint64_t a[4]) { return a[0] == 1 && !a[1] && !a[2] && !a[3]; }
static int geq4(const uint64_t a[4], const uint64_t b[4]) {
    for (int i = 3; i >= 0; i--) {
        if (a[i] > b[i]) return 1;
        if (a[i] < b[i]) return 0;
    }
    return 1;
}
static void sub_mod(uint64_t a[4], const uint64_t b[4]) {
    if (sub4(a, a, b)) add4(a, a, FR_MODULUS);
}

void fr_inverse(fr_t *r, const fr_t *a) {
    if (fr_is_zero(a)) { /* generic_ark.rs:242-245 */
        fr_zero(r);
        return;
    }
    /* treat the Montgomery limbs x = aR as an integer: x^-1 = a^-1 R^-1; multiply by R^3 (montmul) */
    uint64_t u[4], v[4], b[4] = {1, 0, 0, 0}, c[4] = {0, 0, 0, 0};
    memcpy(u, a->l, 32);
    memcpy(v, FR_MODULUS, 32);
    while (!is_one4(u) && !is_one4(v)) {
        while (is_even(u)) {
            shr1(u, 0);
            if (is_even(b)) shr1(b, 0);
            else {
                uint64_t cy = add4(b, b, FR_MODULUS);
                shr1(b, cy);
            }
        }
        while (is_even(v)) {
            shr1(v, 0);
            if (is_even(c)) shr1(c, 0);
            else {
                uint64_t cy = add4(c, c, FR_MODULUS);
                shr1(c, cy);
            }
        }
        if (geq4(u, v)) {
            sub4(u, u, v);
            sub_mod(b, c);
        } else {
            sub4(v, v, u);
            sub_mod(c, b);
        }
    }
    fr_t t;
    memcpy(t.l, is_one4(u) ? b : c, 32);
    fr_mul(r, &t, &R3);
}

void fr_div(fr_t *r, const fr_t *a, const fr_t *b) {
    fr_t inv;
    fr_inverse(&inv, b);
    fr_mul(r, a, &inv);
}

uint32_t fr_num_bits(const fr_t *a) {
    uint64_t c[4];
    fr_to_canonical(a, c);
    for (int i = 3; i >= 0; i--)
        if (c[i]) return (uint32_t)(64 * i + 64 - __builtin_clzll(c[i]));
    return 0;
}
void fr_to_u128(const fr_t *a, uint64_t *lo, uint64_t *hi) {
    uint64_t c[4];
    fr_to_canonical(a, c);
    *lo = c[0];
    *hi = c[1];
}
int fr_try_to_u64(const fr_t *a, uint64_t *v) {
    uint64_t c[4];
    fr_to_canonical(a, c);
    if (c[1] | c[2] | c[3]) return 0;
    *v = c[0];
    return 1;
}
int fr_fetch_nearest_bytes(const fr_t *a, uint32_t num_bits, uint8_t out[32]) {
    uint32_t n = (num_bits + 7) / 8;
    if (n > 32) return -1; /* reference: slice index out of range panic */
    uint64_t c[4];
    fr_to_canonical(a, c);
    for (uint32_t i = 0; i < n; i++) out[i] = (uint8_t)(c[i / 8] >> (8 * (i % 8)));
    return (int)n;
}
static void mask_le(uint8_t le[32], uint32_t num_bits) {
    /* generic_ark.rs:446-473 mask_vector_le, on the little-endian byte view */
    uint32_t mask_power = num_bits % 8, idx = num_bits / 8;
    for (uint32_t i = 0; i < 32; i++) {
        if (i == idx) le[i] &= (uint8_t)((1u << mask_power) - 1);
        else if (i > idx) le[i] = 0;
    }
}
void fr_and_xor(fr_t *r, const fr_t *a, const fr_t *b, uint32_t num_bits, int is_xor) {
    uint8_t x[32], y[32], be[32];
    uint64_t ca[4], cb[4];
    fr_to_canonical(a, ca);
    fr_to_canonical(b, cb);
    for (int i = 0; i < 32; i++) {
        x[i] = (uint8_t)(ca[i / 8] >> (8 * (i % 8)));
        y[i] = (uint8_t)(cb[i / 8] >> (8 * (i % 8)));
    }
    mask_le(x, num_bits);
    mask_le(y, num_bits);
    for (int i = 0; i < 32; i++) be[31 - i] = is_xor ? (x[i] ^ y[i]) : (x[i] & y[i]);
    fr_from_be_bytes_reduce(r, be, 32);
}
void fr_to_hex(const fr_t *a, char out[65]) {
    static const char *hx = "0123456789abcdef";
    uint8_t be[32];
    fr_to_be_bytes(a, be);
    for (int i = 0; i < 32; i++) {
        out[2 * i] = hx[be[i] >> 4];
        out[2 * i + 1] = hx[be[i] & 15];
    }
    out[64] = 0;
}
static int hexval(char c) {
    if (c >= '0' && c <= '9') return c - '0';
    if (c >= 'a' && c <= 'f') return c - 'a' + 10;
    if (c >= 'A' && c <= 'F') return c - 'A' + 10;
    return -1;
}
int fr_from_hex(fr_t *r, const char *s, size_t len) {
    if (len >= 2 && s[0] == '0' && s[1] == 'x') {
        s += 2;
        len -= 2;
    }
    if (len % 2) return -1; /* hex::decode rejects odd length */
    uint8_t buf[256];
    if (len / 2 > sizeof buf) return -1;
    for (size_t i = 0; i < len / 2; i++) {
        int h = hexval(s[2 * i]), l = hexval(s[2 * i + 1]);
        if (h < 0 || l < 0) return -1;
        buf[i] = (uint8_t)(h * 16 + l);
    }
    fr_from_be_bytes_reduce(r, buf, len / 2);
    return 0;
}
