/*
 * oracle/ecdsa.c -- TEST INFRASTRUCTURE ONLY (CPU oracle). Not part of the shipped product path.
 *
 * ECDSA verification over secp256k1 / secp256r1 as the reference calls it:
 *   /root/reference/acvm/src/pwg/blackbox/signature/ecdsa.rs:12-91   (byte extraction, length checks, output)
 *   /root/reference/blackbox_solver/src/lib.rs:66-210               (verify_secp256k1/r1_ecdsa_signature)
 * The arithmetic lives in the third-party crates k256 0.11.6 / p256 0.11.1 (Cargo.lock; not vendored): standard ECDSA
 * (SEC 1 v2 section 4.1.4) restated here with the call-site behaviour of lib.rs:
 *   - Signature::try_from(&[u8; 64]): r and s must be in [1, n-1], else the `.unwrap()` panics          (lib.rs:121,173)
 *   - EncodedPoint::from_affine_coordinates(x, y, compress = true) + PublicKey::from_encoded_point: only x and the PARITY
 *     of y are used; the point is decompressed (y = sqrt(x^3 + ax + b) with that parity); x >= p or a non-residue makes
 *     the `.unwrap()` panic                                                                             (lib.rs:123-128)
 *   - Scalar::from_repr(hashed_msg): the digest must be 32 bytes and < n, else panic                   (lib.rs:130)
 *   - "low S" rule: s > n/2 -> false                                                                    (lib.rs:138-140)
 *   - R = u1*G + u2*Q; the identity is `unreachable!`; Scalar::from_repr(R.x).unwrap() panics if R.x >= n; else R.x == r
 * PINNING: the two vectors of blackbox_solver/src/lib.rs:216-284 (tests/test_oracle_ecdsa.py). The panic message texts are
 * this oracle's wording (the reference's are the crates' Debug strings): parity of the panicking cases is on status,
 * error kind (E_PANIC) and opcode index, and is otherwise unpinned.
 */
#include "pwg.h"
#include <stdio.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } u256;
typedef struct { u256 m, one, r2; uint64_t ninv; } modp_t; /* Montgomery context, R = 2^256 */
typedef struct { modp_t fp, fn; u256 a, b, gx, gy; int a_is_zero; } curve_t; /* a, b, gx, gy in Montgomery form mod p */

static int u_cmp(const u256 *a, const u256 *b) {
    for (int i = 3; i >= 0; i--) { if (a->l[i] < b->l[i]) return -1; if (a->l[i] > b->l[i]) return 1; }
    return 0;
}
static int u_is_zero(const u256 *a) { return !(a->l[0] | a->l[1] | a->l[2] | a->l[3]); }
static uint64_t u_add(u256 *r, const u256 *a, const u256 *b) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a->l[i] + b->l[i]; r->l[i] = (uint64_t)c; c >>= 64; }
    return (uint64_t)c;
}
static uint64_t u_sub(u256 *r, const u256 *a, const u256 *b) {
    uint64_t borrow = 0;
    for (int i = 0; i < 4; i++) { u128 d = (u128)a->l[i] - b->l[i] - borrow; r->l[i] = (uint64_t)d; borrow = (uint64_t)(d >> 64) & 1; }
    return borrow;
}
static void from_be(u256 *r, const uint8_t be[32]) {
    memset(r, 0, sizeof *r);
    for (int i = 0; i < 32; i++) r->l[i / 8] |= (uint64_t)be[31 - i] << (8 * (i % 8));
}
static void from_hex(u256 *r, const char *h) { /* 64 hex digits */
    uint8_t be[32];
    for (int i = 0; i < 32; i++) { unsigned v; sscanf(h + 2 * i, "%2x", &v); be[i] = (uint8_t)v; }
    from_be(r, be);
}
/* generic Montgomery product for a full 256-bit odd modulus (CIOS with the extra carry word) */
static void m_mul(u256 *r, const u256 *a, const u256 *b, const modp_t *f) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)a->l[j] * b->l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * f->ninv;
        c = (u128)m * f->m.l[0] + t[0]; c >>= 64;
        for (int j = 1; j < 4; j++) { c += (u128)m * f->m.l[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    u256 v = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || u_cmp(&v, &f->m) >= 0) u_sub(&v, &v, &f->m);
    *r = v;
}
static void m_add(u256 *r, const u256 *a, const u256 *b, const modp_t *f) {
    uint64_t c = u_add(r, a, b);
    if (c || u_cmp(r, &f->m) >= 0) u_sub(r, r, &f->m);
}
static void m_sub(u256 *r, const u256 *a, const u256 *b, const modp_t *f) {
    if (u_sub(r, a, b)) u_add(r, r, &f->m);
}
static void m_to(u256 *r, const u256 *a, const modp_t *f) { m_mul(r, a, &f->r2, f); }
static void m_from(u256 *r, const u256 *a, const modp_t *f) { u256 o = {{1, 0, 0, 0}}; m_mul(r, a, &o, f); }
static void m_pow(u256 *r, const u256 *a, const u256 *e, const modp_t *f) {
    u256 acc = f->one;
    for (int i = 255; i >= 0; i--) {
        m_mul(&acc, &acc, &acc, f);
        if ((e->l[i / 64] >> (i % 64)) & 1) m_mul(&acc, &acc, a, f);
    }
    *r = acc;
}
static void m_inv(u256 *r, const u256 *a, const modp_t *f) { /* prime modulus: a^(m-2) */
    u256 e = f->m, two = {{2, 0, 0, 0}};
    u_sub(&e, &e, &two);
    m_pow(r, a, &e, f);
}
static void ctx_init(modp_t *f, const char *mod_hex) {
    from_hex(&f->m, mod_hex);
    uint64_t inv = 1;
    for (int i = 0; i < 6; i++) inv *= 2 - f->m.l[0] * inv; /* m^-1 mod 2^64 */
    f->ninv = 0 - inv;
    /* one = 2^256 mod m, r2 = 2^512 mod m by doubling */
    u256 x = {{1, 0, 0, 0}};
    for (int i = 0; i < 512; i++) {
        uint64_t c = u_add(&x, &x, &x);
        if (c || u_cmp(&x, &f->m) >= 0) u_sub(&x, &x, &f->m);
        if (i == 255) f->one = x;
    }
    f->r2 = x;
}

static curve_t K1, R1;
static int g_init;
static void curves_init(void) {
    if (g_init) return;
    ctx_init(&K1.fp, "FFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEFFFFFC2F");
    ctx_init(&K1.fn, "FFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141");
    ctx_init(&R1.fp, "FFFFFFFF00000001000000000000000000000000FFFFFFFFFFFFFFFFFFFFFFFF");
    ctx_init(&R1.fn, "FFFFFFFF00000000FFFFFFFFFFFFFFFFBCE6FAADA7179E84F3B9CAC2FC632551");
    u256 t;
    memset(&K1.a, 0, sizeof K1.a);
    K1.a_is_zero = 1;
    t = (u256){{7, 0, 0, 0}}; m_to(&K1.b, &t, &K1.fp);
    from_hex(&t, "79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798"); m_to(&K1.gx, &t, &K1.fp);
    from_hex(&t, "483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8"); m_to(&K1.gy, &t, &K1.fp);
    t = (u256){{3, 0, 0, 0}}; m_to(&t, &t, &R1.fp);
    u256 z = {{0, 0, 0, 0}};
    m_sub(&R1.a, &z, &t, &R1.fp); /* a = -3 */
    R1.a_is_zero = 0;
    from_hex(&t, "5AC635D8AA3A93E7B3EBBD55769886BC651D06B0CC53B0F63BCE3C3E27D2604B"); m_to(&R1.b, &t, &R1.fp);
    from_hex(&t, "6B17D1F2E12C4247F8BCE6E563A440F277037D812DEB33A0F4A13945D898C296"); m_to(&R1.gx, &t, &R1.fp);
    from_hex(&t, "4FE342E2FE1A7F9B8EE7EB4A7C0F9E162BCE33576B315ECECBB6406837BF51F5"); m_to(&R1.gy, &t, &R1.fp);
    g_init = 1;
}

typedef struct { u256 X, Y, Z; } jac_t; /* Montgomery coordinates, Z == 0 <=> identity */
static void jac_dbl(jac_t *r, const jac_t *p, const curve_t *c) {
    const modp_t *f = &c->fp;
    if (u_is_zero(&p->Z) || u_is_zero(&p->Y)) { memset(r, 0, sizeof *r); r->X = f->one; r->Y = f->one; return; }
    u256 xx, yy, yyyy, zz, s, m, t, X3, Y3, Z3;
    m_mul(&xx, &p->X, &p->X, f); m_mul(&yy, &p->Y, &p->Y, f); m_mul(&yyyy, &yy, &yy, f); m_mul(&zz, &p->Z, &p->Z, f);
    m_mul(&s, &p->X, &yy, f); m_add(&s, &s, &s, f); m_add(&s, &s, &s, f);           /* S = 4 X Y^2 */
    m_add(&m, &xx, &xx, f); m_add(&m, &m, &xx, f);                                     /* M = 3 X^2 + a Z^4 */
    if (!c->a_is_zero) { m_mul(&t, &zz, &zz, f); m_mul(&t, &t, &c->a, f); m_add(&m, &m, &t, f); }
    m_mul(&X3, &m, &m, f); m_sub(&X3, &X3, &s, f); m_sub(&X3, &X3, &s, f);
    m_sub(&t, &s, &X3, f); m_mul(&Y3, &m, &t, f);
    m_add(&t, &yyyy, &yyyy, f); m_add(&t, &t, &t, f); m_add(&t, &t, &t, f); m_sub(&Y3, &Y3, &t, f);
    m_mul(&Z3, &p->Y, &p->Z, f); m_add(&Z3, &Z3, &Z3, f);
    r->X = X3; r->Y = Y3; r->Z = Z3;
}
static void jac_add(jac_t *r, const jac_t *p, const jac_t *q, const curve_t *c) {
    const modp_t *f = &c->fp;
    if (u_is_zero(&p->Z)) { *r = *q; return; }
    if (u_is_zero(&q->Z)) { *r = *p; return; }
    u256 z1z1, z2z2, u1, u2, s1, s2, h, rr, hh, hhh, v, t, X3, Y3, Z3;
    m_mul(&z1z1, &p->Z, &p->Z, f); m_mul(&z2z2, &q->Z, &q->Z, f);
    m_mul(&u1, &p->X, &z2z2, f); m_mul(&u2, &q->X, &z1z1, f);
    m_mul(&s1, &p->Y, &q->Z, f); m_mul(&s1, &s1, &z2z2, f);
    m_mul(&s2, &q->Y, &p->Z, f); m_mul(&s2, &s2, &z1z1, f);
    m_sub(&h, &u2, &u1, f); m_sub(&rr, &s2, &s1, f);
    if (u_is_zero(&h)) {
        if (u_is_zero(&rr)) { jac_dbl(r, p, c); return; }
        memset(r, 0, sizeof *r); r->X = f->one; r->Y = f->one; return;
    }
    m_mul(&hh, &h, &h, f); m_mul(&hhh, &hh, &h, f); m_mul(&v, &u1, &hh, f);
    m_mul(&X3, &rr, &rr, f); m_sub(&X3, &X3, &hhh, f); m_sub(&X3, &X3, &v, f); m_sub(&X3, &X3, &v, f);
    m_sub(&t, &v, &X3, f); m_mul(&Y3, &rr, &t, f); m_mul(&t, &s1, &hhh, f); m_sub(&Y3, &Y3, &t, f);
    m_mul(&Z3, &p->Z, &q->Z, f); m_mul(&Z3, &Z3, &h, f);
    r->X = X3; r->Y = Y3; r->Z = Z3;
}
static void jac_mul(jac_t *r, const u256 *x, const u256 *y, const u256 *k, const curve_t *c) { /* k plain integer */
    jac_t acc, p;
    memset(&acc, 0, sizeof acc); acc.X = c->fp.one; acc.Y = c->fp.one;
    p.X = *x; p.Y = *y; p.Z = c->fp.one;
    for (int i = 255; i >= 0; i--) {
        jac_dbl(&acc, &acc, c);
        if ((k->l[i / 64] >> (i % 64)) & 1) jac_add(&acc, &acc, &p, c);
    }
    *r = acc;
}

/* returns 0 / 1 (invalid / valid) or a negative panic code; curve 0 = secp256k1, 1 = secp256r1 */
enum { ECDSA_PANIC_SIG = -1, ECDSA_PANIC_PUBKEY = -2, ECDSA_PANIC_MSG_LEN = -3, ECDSA_PANIC_MSG_RANGE = -4, ECDSA_PANIC_IDENTITY = -5, ECDSA_PANIC_X_RANGE = -6 };
int oracle_ecdsa_verify(int curve, const uint8_t *hashed_msg, size_t msg_len, const uint8_t pkx[32], const uint8_t pky[32], const uint8_t sig[64]) {
    curves_init();
    const curve_t *c = curve ? &R1 : &K1;
    const modp_t *fp = &c->fp, *fn = &c->fn;
    u256 r, s, x, z, t;
    from_be(&r, sig);
    from_be(&s, sig + 32);
    if (u_is_zero(&r) || u_is_zero(&s) || u_cmp(&r, &fn->m) >= 0 || u_cmp(&s, &fn->m) >= 0) return ECDSA_PANIC_SIG;
    from_be(&x, pkx);
    if (u_cmp(&x, &fp->m) >= 0) return ECDSA_PANIC_PUBKEY;
    u256 xm, rhs, ym, e;
    m_to(&xm, &x, fp);
    m_mul(&rhs, &xm, &xm, fp); m_mul(&rhs, &rhs, &xm, fp);
    if (!c->a_is_zero) { m_mul(&t, &c->a, &xm, fp); m_add(&rhs, &rhs, &t, fp); }
    m_add(&rhs, &rhs, &c->b, fp);
    /* p = 3 mod 4 on both curves: sqrt = rhs^((p+1)/4) */
    e = fp->m;
    u256 one_i = {{1, 0, 0, 0}};
    u_add(&e, &e, &one_i); /* p + 1 overflows only to 2^256 for neither curve (p + 1 < 2^256) */
    for (int i = 0; i < 4; i++) e.l[i] = (e.l[i] >> 2) | (i < 3 ? e.l[i + 1] << 62 : 0);
    m_pow(&ym, &rhs, &e, fp);
    m_mul(&t, &ym, &ym, fp);
    if (u_cmp(&t, &rhs) != 0) return ECDSA_PANIC_PUBKEY;
    u256 yc;
    m_from(&yc, &ym, fp);
    if ((yc.l[0] & 1) != (uint64_t)(pky[31] & 1)) { u256 zero = {{0, 0, 0, 0}}; m_sub(&ym, &zero, &ym, fp); }
    if (msg_len != 32) return ECDSA_PANIC_MSG_LEN;
    from_be(&z, hashed_msg);
    if (u_cmp(&z, &fn->m) >= 0) return ECDSA_PANIC_MSG_RANGE;
    /* low-S: s > n / 2 -> false */
    u256 half;
    for (int i = 0; i < 4; i++) half.l[i] = (fn->m.l[i] >> 1) | (i < 3 ? fn->m.l[i + 1] << 63 : 0);
    if (u_cmp(&s, &half) > 0) return 0;
    u256 sm, si, zm, rm, u1, u2;
    m_to(&sm, &s, fn); m_inv(&si, &sm, fn);
    m_to(&zm, &z, fn); m_to(&rm, &r, fn);
    m_mul(&u1, &zm, &si, fn); m_mul(&u2, &rm, &si, fn);
    m_from(&u1, &u1, fn); m_from(&u2, &u2, fn);
    jac_t a, b, R;
    jac_mul(&a, &c->gx, &c->gy, &u1, c);
    jac_mul(&b, &xm, &ym, &u2, c);
    jac_add(&R, &a, &b, c);
    if (u_is_zero(&R.Z)) return ECDSA_PANIC_IDENTITY;
    u256 zi, zi2, rx;
    m_inv(&zi, &R.Z, fp); m_mul(&zi2, &zi, &zi, fp); m_mul(&rx, &R.X, &zi2, fp); m_from(&rx, &rx, fp);
    if (u_cmp(&rx, &fn->m) >= 0) return ECDSA_PANIC_X_RANGE;
    return u_cmp(&rx, &r) == 0;
}
const char *oracle_ecdsa_panic_text(int code) {
    switch (code) {
    case ECDSA_PANIC_SIG: return "ecdsa: signature scalars must be in [1, n-1] (Signature::try_from unwrap)";
    case ECDSA_PANIC_PUBKEY: return "ecdsa: public key x is not on the curve (PublicKey::from_encoded_point unwrap)";
    case ECDSA_PANIC_MSG_LEN: return "ecdsa: hashed message must be 32 bytes (GenericArray::from_slice)";
    case ECDSA_PANIC_MSG_RANGE: return "ecdsa: hashed message is not below the group order (Scalar::from_repr unwrap)";
    case ECDSA_PANIC_IDENTITY: return "ecdsa: R is the identity (unreachable!)";
    case ECDSA_PANIC_X_RANGE: return "ecdsa: R.x is not below the group order (Scalar::from_repr unwrap)";
    }
    return "";
}
