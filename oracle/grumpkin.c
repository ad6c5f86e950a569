/*
 * oracle/grumpkin.c -- TEST INFRASTRUCTURE ONLY (CPU oracle). Not part of the shipped product path.
 *
 * The three BlackBoxFunctionSolver functions the reference delegates to barretenberg's acvm_backend.wasm
 * (AztecProtocol/barretenberg v0.5.0 per barretenberg_blackbox_solver/build.rs:10; un-vendored, absent here):
 *   fixed_base_scalar_mul  <- /root/reference/barretenberg_blackbox_solver/src/wasm/scalar_mul.rs:17-65
 *                             (limb / modulus checks are in-tree; `compute_public_key` is the wasm export)
 *   pedersen               <- .../wasm/pedersen.rs:14-35 (`pedersen_plookup_commit_with_hash_index`)
 *   schnorr_verify         <- .../src/lib.rs:40-58, wasm/schnorr.rs:68-103 (`verify_signature`)
 * The arithmetic restates the published barretenberg algorithms as reconstructed and checked in SURVEY.md
 * Appendix A. PINNING: all five golden vectors the reference holds are reproduced by tests/test_oracle_grumpkin.py
 * (scalar_mul.rs:72-97 x2, pedersen.rs:38-54, acvm_js/test/shared/pedersen.ts, schnorr_verify.ts).
 * PARITY UNPINNED (no reference vector exists): Pedersen hash_index != 0, Schnorr early rejects / rejecting
 * signatures, the encoding of the point at infinity (scalar 0, empty Pedersen input).
 */
#include "hashes.h"
#include "pwg.h"
#include <pthread.h>
#include <stdio.h>
#include <string.h>

typedef struct { fr_t x, y; int inf; } aff_t;
typedef struct { fr_t X, Y, Z; } jac_t; /* Z == 0 <=> infinity */

/* group order q = BN254 Fq (scalar_mul.rs:42-45), little-endian limbs */
static const uint64_t GQ[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};

static fr_t B_COEF;      /* -17 */
static fr_t BETA;        /* cube root of unity, SURVEY A.2 */
static aff_t GEN_ONE;    /* G = (1, sqrt(-16)) */
#define N_GENERATORS 30
static aff_t GENS[N_GENERATORS];
static pthread_once_t g_once = PTHREAD_ONCE_INIT;

/* ---------------------------------------------------------------- field helpers */
static void fr_pow(fr_t *r, const fr_t *a, const uint64_t e[4]) {
    fr_t acc;
    fr_one(&acc);
    for (int i = 255; i >= 0; i--) {
        fr_mul(&acc, &acc, &acc);
        if ((e[i / 64] >> (i % 64)) & 1) fr_mul(&acc, &acc, a);
    }
    *r = acc;
}
/* Tonelli-Shanks over Fr (p - 1 = 2^28 * t). Returns 0 if a is a non-residue. */
static int fr_sqrt(fr_t *r, const fr_t *a) {
    if (fr_is_zero(a)) { fr_zero(r); return 1; }
    uint64_t pm1[4];
    memcpy(pm1, FR_MODULUS, 32);
    pm1[0] -= 1;
    /* t = (p-1) >> 28 ; (t+1)/2 ; (p-1)/2 */
    uint64_t t[4], t1h[4], half[4];
    for (int i = 0; i < 4; i++) t[i] = (pm1[i] >> 28) | (i < 3 ? pm1[i + 1] << 36 : 0);
    for (int i = 0; i < 4; i++) half[i] = (pm1[i] >> 1) | (i < 3 ? pm1[i + 1] << 63 : 0);
    uint64_t tp1[4] = {t[0] + 1, t[1], t[2], t[3]}; /* t is odd: no carry */
    for (int i = 0; i < 4; i++) t1h[i] = (tp1[i] >> 1) | (i < 3 ? tp1[i + 1] << 63 : 0);
    fr_t ls;
    fr_pow(&ls, a, half);
    if (!fr_is_one(&ls)) return 0;
    fr_t g5, z, c, x, b;
    fr_from_u64(&g5, 5); /* 5 generates Fr^* for BN254 */
    fr_pow(&z, &g5, t);
    c = z;
    fr_pow(&x, a, t1h);
    fr_pow(&b, a, t);
    int m = 28;
    while (!fr_is_one(&b)) {
        int i = 0;
        fr_t b2 = b;
        while (!fr_is_one(&b2)) { fr_mul(&b2, &b2, &b2); i++; }
        fr_t e = c;
        for (int k = 0; k < m - i - 1; k++) fr_mul(&e, &e, &e);
        fr_mul(&x, &x, &e);
        fr_mul(&c, &e, &e);
        fr_mul(&b, &b, &c);
        m = i;
    }
    *r = x;
    return 1;
}

/* ---------------------------------------------------------------- curve y^2 = x^3 - 17 */
static int on_curve(const aff_t *p) {
    if (p->inf) return 1;
    fr_t l, r;
    fr_mul(&l, &p->y, &p->y);
    fr_mul(&r, &p->x, &p->x);
    fr_mul(&r, &r, &p->x);
    fr_add(&r, &r, &B_COEF);
    return fr_eq(&l, &r);
}
static void jac_inf(jac_t *r) { fr_one(&r->X); fr_one(&r->Y); fr_zero(&r->Z); }
static int jac_is_inf(const jac_t *p) { return fr_is_zero(&p->Z); }
static void jac_from_aff(jac_t *r, const aff_t *p) {
    if (p->inf) { jac_inf(r); return; }
    r->X = p->x; r->Y = p->y; fr_one(&r->Z);
}
static void jac_dbl(jac_t *r, const jac_t *p) {
    if (jac_is_inf(p) || fr_is_zero(&p->Y)) { jac_inf(r); return; }
    fr_t A, B, C, D, E, F, t, X3, Y3, Z3;
    fr_mul(&A, &p->X, &p->X);
    fr_mul(&B, &p->Y, &p->Y);
    fr_mul(&C, &B, &B);
    fr_add(&t, &p->X, &B); fr_mul(&t, &t, &t); fr_sub(&t, &t, &A); fr_sub(&t, &t, &C);
    fr_add(&D, &t, &t);
    fr_add(&E, &A, &A); fr_add(&E, &E, &A);
    fr_mul(&F, &E, &E);
    fr_sub(&X3, &F, &D); fr_sub(&X3, &X3, &D);
    fr_sub(&t, &D, &X3); fr_mul(&Y3, &E, &t);
    fr_add(&t, &C, &C); fr_add(&t, &t, &t); fr_add(&t, &t, &t);
    fr_sub(&Y3, &Y3, &t);
    fr_mul(&Z3, &p->Y, &p->Z); fr_add(&Z3, &Z3, &Z3);
    r->X = X3; r->Y = Y3; r->Z = Z3;
}
static void jac_add(jac_t *r, const jac_t *p, const jac_t *q) {
    if (jac_is_inf(p)) { *r = *q; return; }
    if (jac_is_inf(q)) { *r = *p; return; }
    fr_t Z1Z1, Z2Z2, U1, U2, S1, S2, H, rr, I, J, V, t, X3, Y3, Z3;
    fr_mul(&Z1Z1, &p->Z, &p->Z);
    fr_mul(&Z2Z2, &q->Z, &q->Z);
    fr_mul(&U1, &p->X, &Z2Z2);
    fr_mul(&U2, &q->X, &Z1Z1);
    fr_mul(&S1, &p->Y, &q->Z); fr_mul(&S1, &S1, &Z2Z2);
    fr_mul(&S2, &q->Y, &p->Z); fr_mul(&S2, &S2, &Z1Z1);
    fr_sub(&H, &U2, &U1);
    fr_sub(&rr, &S2, &S1);
    if (fr_is_zero(&H)) {
        if (fr_is_zero(&rr)) jac_dbl(r, p);
        else jac_inf(r);
        return;
    }
    fr_add(&I, &H, &H); fr_mul(&I, &I, &I);
    fr_mul(&J, &H, &I);
    fr_add(&rr, &rr, &rr);
    fr_mul(&V, &U1, &I);
    fr_mul(&X3, &rr, &rr); fr_sub(&X3, &X3, &J); fr_sub(&X3, &X3, &V); fr_sub(&X3, &X3, &V);
    fr_sub(&t, &V, &X3); fr_mul(&Y3, &rr, &t);
    fr_mul(&t, &S1, &J); fr_add(&t, &t, &t); fr_sub(&Y3, &Y3, &t);
    fr_add(&Z3, &p->Z, &q->Z); fr_mul(&Z3, &Z3, &Z3); fr_sub(&Z3, &Z3, &Z1Z1); fr_sub(&Z3, &Z3, &Z2Z2);
    fr_mul(&Z3, &Z3, &H);
    r->X = X3; r->Y = Y3; r->Z = Z3;
}
static void jac_neg(jac_t *r, const jac_t *p) { *r = *p; fr_neg(&r->Y, &p->Y); }
static void jac_to_aff(aff_t *r, const jac_t *p) {
    if (jac_is_inf(p)) { fr_zero(&r->x); fr_zero(&r->y); r->inf = 1; return; }
    fr_t zi, zi2, zi3;
    fr_inverse(&zi, &p->Z);
    fr_mul(&zi2, &zi, &zi);
    fr_mul(&zi3, &zi2, &zi);
    fr_mul(&r->x, &p->X, &zi2);
    fr_mul(&r->y, &p->Y, &zi3);
    r->inf = 0;
}
/* k * P for a 256-bit little-endian-limb integer k (plain double-and-add, MSB first) */
static void jac_mul(jac_t *r, const aff_t *p, const uint64_t k[4]) {
    jac_t acc, pj;
    jac_inf(&acc);
    jac_from_aff(&pj, p);
    for (int i = 255; i >= 0; i--) {
        jac_dbl(&acc, &acc);
        if ((k[i / 64] >> (i % 64)) & 1) jac_add(&acc, &acc, &pj);
    }
    *r = acc;
}

/* ---------------------------------------------------------------- generators (SURVEY A.2 derive_generators) */
static void init_once(void) {
    fr_t seventeen;
    fr_from_u64(&seventeen, 17);
    fr_neg(&B_COEF, &seventeen);
    static const uint8_t beta_be[32] = {0, 0, 0, 0, 0, 0, 0, 0, 0xb3, 0xc4, 0xd7, 0x9d, 0x41, 0xa9, 0x17, 0x58,
                                        0x5b, 0xfc, 0x41, 0x08, 0x8d, 0x8d, 0xaa, 0xa7, 0x8b, 0x17, 0xea, 0x66, 0xb9, 0x9c, 0x90, 0xdd};
    fr_from_be_bytes_reduce(&BETA, beta_be, 32);
    /* G = (1, y) with y^2 = 1 - 17 = -16, y = 0x...02cf135e7506a45d632d270d45f1181294833fc48d823f272c (scalar_mul.rs:77-78) */
    static const uint8_t gy_be[32] = {0, 0, 0, 0, 0, 0, 0, 0x02, 0xcf, 0x13, 0x5e, 0x75, 0x06, 0xa4, 0x5d, 0x63,
                                      0x2d, 0x27, 0x0d, 0x45, 0xf1, 0x18, 0x12, 0x94, 0x83, 0x3f, 0xc4, 0x8d, 0x82, 0x3f, 0x27, 0x2c};
    fr_one(&GEN_ONE.x);
    fr_from_be_bytes_reduce(&GEN_ONE.y, gy_be, 32);
    GEN_ONE.inf = 0;
    int found = 0;
    for (uint64_t seed = 1; found < N_GENERATORS; seed++) {
        uint8_t buf[32], h[32], le_rev[32];
        memset(buf, 0, 32);
        for (int i = 0; i < 8; i++) buf[i] = (uint8_t)(seed >> (8 * (7 - i)));
        oracle_keccak256(buf, 32, h);
        /* digest bytes as a LITTLE-endian integer; y_bit = bit 255; x = (hv mod 2^255) mod p */
        int y_bit = h[31] >> 7;
        for (int i = 0; i < 32; i++) le_rev[i] = h[31 - i];
        le_rev[0] &= 0x7f;
        fr_t x, yy, y;
        fr_from_be_bytes_reduce(&x, le_rev, 32);
        fr_mul(&yy, &x, &x);
        fr_mul(&yy, &yy, &x);
        fr_add(&yy, &yy, &B_COEF);
        if (!fr_sqrt(&y, &yy)) continue;
        uint64_t yc[4];
        fr_to_canonical(&y, yc);
        if ((int)(yc[0] & 1) != y_bit) fr_neg(&y, &y);
        GENS[found].x = x;
        GENS[found].y = y;
        GENS[found].inf = 0;
        found++;
    }
}
static void ensure_init(void) { pthread_once(&g_once, init_once); }

/* ---------------------------------------------------------------- fixed base */
static void hex32(const fr_t *v, char out[65]) { fr_to_hex(v, out); }

static int bb_fixed_base(void *ctx, const fr_t *low, const fr_t *high, fr_t *x, fr_t *y, char *err, size_t errlen) {
    (void)ctx;
    ensure_init();
    char hx[65];
    if (fr_num_bits(low) > 128) { /* scalar_mul.rs:25-29 try_into_u128 */
        hex32(low, hx);
        snprintf(err, errlen, "Limb %s is not less than 2^128", hx);
        return 1;
    }
    if (fr_num_bits(high) > 128) {
        hex32(high, hx);
        snprintf(err, errlen, "Limb %s is not less than 2^128", hx);
        return 1;
    }
    uint64_t lo[4], hi[4], k[4];
    fr_to_canonical(low, lo);
    fr_to_canonical(high, hi);
    k[0] = lo[0]; k[1] = lo[1]; k[2] = hi[0]; k[3] = hi[1];
    int ge = 1;
    for (int i = 3; i >= 0; i--) {
        if (k[i] > GQ[i]) break;
        if (k[i] < GQ[i]) { ge = 0; break; }
    }
    if (ge) { /* scalar_mul.rs:41-51; hex::encode(BigUint::to_bytes_be): minimal big-endian bytes */
        uint8_t be[32];
        for (int i = 0; i < 32; i++) be[31 - i] = (uint8_t)(k[i / 8] >> (8 * (i % 8)));
        int s = 0;
        while (s < 31 && be[s] == 0) s++;
        char hexs[65];
        for (int i = s; i < 32; i++) snprintf(hexs + 2 * (i - s), 3, "%02x", be[i]);
        snprintf(err, errlen, "Value %s is not a valid grumpkin scalar", hexs);
        return 1;
    }
    jac_t r;
    aff_t a;
    jac_mul(&r, &GEN_ONE, k);
    jac_to_aff(&a, &r);
    *x = a.x; /* infinity (scalar 0) -> (0,0): encoding unpinned */
    *y = a.y;
    return 0;
}

/* ---------------------------------------------------------------- plookup pedersen */
static void small_mul(jac_t *r, const aff_t *g, uint32_t k) { /* k in 1..512 */
    uint64_t kk[4] = {k, 0, 0, 0};
    jac_t acc, pj;
    jac_inf(&acc);
    jac_from_aff(&pj, g);
    for (int i = 9; i >= 0; i--) {
        jac_dbl(&acc, &acc);
        if ((kk[0] >> i) & 1) jac_add(&acc, &acc, &pj);
    }
    *r = acc;
}
static void hash_single(jac_t *out, const fr_t *v, int parity) {
    uint64_t bits[4];
    fr_to_canonical(v, bits);
    int off = parity ? 15 : 0;
    jac_t acc0, acc1, t;
    jac_inf(&acc0);
    jac_inf(&acc1);
    unsigned pos = 0;
    for (int i = 0; i < 15; i++) {
        uint32_t a = 0, b = 0;
        for (int k = 0; k < 9; k++, pos++) if (pos < 256) a |= (uint32_t)((bits[pos / 64] >> (pos % 64)) & 1) << k;
        for (int k = 0; k < 9; k++, pos++) if (pos < 256) b |= (uint32_t)((bits[pos / 64] >> (pos % 64)) & 1) << k;
        small_mul(&t, &GENS[off + i], a + 1);
        jac_add(&acc0, &acc0, &t);
        if (i < 14) {
            small_mul(&t, &GENS[off + i], b + 1);
            jac_add(&acc1, &acc1, &t);
        }
    }
    /* endomorphism on the first accumulator: (x, y) -> (beta * x, y); in Jacobian coordinates X -> beta * X */
    fr_mul(&acc0.X, &acc0.X, &BETA);
    jac_add(out, &acc0, &acc1);
}
static void hash_pair_x(fr_t *out, const fr_t *l, const fr_t *r) {
    jac_t a, b, s;
    aff_t p;
    hash_single(&a, l, 0);
    hash_single(&b, r, 1);
    jac_add(&s, &a, &b);
    jac_to_aff(&p, &s);
    *out = p.x;
}
static int bb_pedersen(void *ctx, const fr_t *inputs, size_t n, uint32_t hash_index, fr_t *x, fr_t *y, char *err, size_t errlen) {
    (void)ctx; (void)err; (void)errlen;
    ensure_init();
    if (n == 0) { fr_zero(x); fr_zero(y); return 0; } /* point at infinity: encoding unpinned */
    /* IV[hash_index] = (hash_index + 1) * G (recollection, unpinned for hash_index != 0); IV[0].x = 1 */
    fr_t r, nf;
    if (hash_index == 0) fr_one(&r);
    else {
        uint64_t k[4] = {(uint64_t)hash_index + 1, 0, 0, 0};
        jac_t ivj;
        aff_t iv;
        jac_mul(&ivj, &GEN_ONE, k);
        jac_to_aff(&iv, &ivj);
        r = iv.x;
    }
    fr_from_u64(&nf, (uint64_t)n);
    hash_pair_x(&r, &r, &nf);
    for (size_t i = 0; i + 1 < n; i++) hash_pair_x(&r, &r, &inputs[i]);
    jac_t a, b, s;
    aff_t p;
    hash_single(&a, &r, 0);
    hash_single(&b, &inputs[n - 1], 1);
    jac_add(&s, &a, &b);
    jac_to_aff(&p, &s);
    *x = p.x;
    *y = p.y;
    return 0;
}

/* ---------------------------------------------------------------- schnorr */
/* H_j(v) of the non-lookup "hash ladder" pedersen: generators (g, aux, skew) = D[3j], D[3j+1], D[3j+2] */
static void ladder_term(jac_t *out, const fr_t *v, int j) {
    uint64_t V[4];
    fr_to_canonical(v, V);
    int even = !(V[0] & 1);
    if (even) { /* V = v + 1 (v < p: no overflow of 256 bits) */
        for (int i = 0; i < 4; i++) { if (++V[i]) break; }
    }
    int t = (int)((V[0] + 16) & 31);
    int lo = t < 16 ? t : t - 32; /* odd, -15..15 */
    /* hi = (V - lo) / 16 */
    uint64_t hi[4];
    memcpy(hi, V, 32);
    if (lo >= 0) {
        uint64_t borrow = (uint64_t)lo;
        for (int i = 0; i < 4 && borrow; i++) { uint64_t o = hi[i]; hi[i] -= borrow; borrow = hi[i] > o; }
    } else {
        uint64_t carry = (uint64_t)(-lo);
        for (int i = 0; i < 4 && carry; i++) { uint64_t o = hi[i]; hi[i] += carry; carry = hi[i] < o; }
    }
    for (int i = 0; i < 4; i++) hi[i] = (hi[i] >> 4) | (i < 3 ? hi[i + 1] << 60 : 0);
    jac_t p, q;
    jac_mul(&p, &GENS[3 * j], hi);
    uint64_t al[4] = {(uint64_t)(lo < 0 ? -lo : lo), 0, 0, 0};
    jac_mul(&q, &GENS[3 * j + 1], al);
    if (lo < 0) jac_neg(&q, &q);
    jac_add(&p, &p, &q);
    if (even) {
        jac_t sk;
        jac_from_aff(&sk, &GENS[3 * j + 2]);
        jac_neg(&sk, &sk);
        jac_add(&p, &p, &sk);
    }
    *out = p;
}
static void compress(fr_t *out, const fr_t *v, int m) {
    jac_t acc, t;
    aff_t a;
    jac_inf(&acc);
    for (int j = 0; j < m; j++) {
        ladder_term(&t, &v[j], j);
        jac_add(&acc, &acc, &t);
    }
    jac_to_aff(&a, &acc);
    *out = a.x;
}
static void reduce_mod_q(uint64_t k[4]) { /* k < 2^256 < 6q */
    for (;;) {
        int ge = 1;
        for (int i = 3; i >= 0; i--) {
            if (k[i] > GQ[i]) break;
            if (k[i] < GQ[i]) { ge = 0; break; }
        }
        if (!ge) return;
        uint64_t borrow = 0;
        for (int i = 0; i < 4; i++) {
            unsigned __int128 d = (unsigned __int128)k[i] - GQ[i] - borrow;
            k[i] = (uint64_t)d;
            borrow = (uint64_t)(d >> 64) & 1;
        }
    }
}
static void be_to_limbs(const uint8_t be[32], uint64_t k[4]) {
    memset(k, 0, 32);
    for (int i = 0; i < 32; i++) k[i / 8] |= (uint64_t)be[31 - i] << (8 * (i % 8));
}
static void challenge(uint8_t out[32], const fr_t *rx, const fr_t *pkx, const fr_t *pky, const uint8_t *msg, size_t msg_len) {
    fr_t v[3] = {*rx, *pkx, *pky}, c;
    compress(&c, v, 3);
    uint8_t buf[32 + 1024];
    fr_to_be_bytes(&c, buf);
    memcpy(buf + 32, msg, msg_len);
    oracle_blake2s(buf, 32 + msg_len, out);
}
static int bb_schnorr(void *ctx, const fr_t *pkx, const fr_t *pky, const uint8_t *sig, size_t sig_len, const uint8_t *msg,
                      size_t msg_len, int *ok, char *err, size_t errlen) {
    (void)ctx;
    ensure_init();
    if (sig_len < 64) { /* lib.rs:50-52: signature[0..32] / [32..64] slicing panics */
        snprintf(err, errlen, "range end index 64 out of range for slice of length %zu", sig_len);
        return 3;
    }
    if (128 + msg_len >= 1024) { /* wasm/schnorr.rs:79-82 */
        snprintf(err, errlen, "Message overran wasm scratch space");
        return 3;
    }
    *ok = 0;
    aff_t pk = {*pkx, *pky, 0};
    uint64_t s[4], e[4];
    be_to_limbs(sig, s);
    be_to_limbs(sig + 32, e);
    reduce_mod_q(s);
    reduce_mod_q(e);
    /* early rejects: recollection of barretenberg, not pinned by any reference vector */
    if (!on_curve(&pk)) return 0;
    if (!(s[0] | s[1] | s[2] | s[3]) || !(e[0] | e[1] | e[2] | e[3])) return 0;
    jac_t a, b, r;
    aff_t R;
    jac_mul(&a, &pk, e);
    jac_mul(&b, &GEN_ONE, s);
    jac_add(&r, &a, &b);
    if (jac_is_inf(&r)) return 0;
    jac_to_aff(&R, &r);
    uint8_t target[32];
    challenge(target, &R.x, pkx, pky, msg, msg_len);
    *ok = memcmp(target, sig + 32, 32) == 0;
    return 0;
}

const backend_t ORACLE_BARRETENBERG_BACKEND = {0, bb_schnorr, bb_pedersen, bb_fixed_base};

/* ---------------------------------------------------------------- exported helpers for tests / input generation */
/* generator D[i] (i < 30) as x||y big-endian */
void oracle_grumpkin_generator(uint32_t i, uint8_t out[64]) {
    ensure_init();
    fr_to_be_bytes(&GENS[i].x, out);
    fr_to_be_bytes(&GENS[i].y, out + 32);
}
/* k * G for a 32-byte big-endian integer k; returns 1 if the result is the point at infinity */
int oracle_grumpkin_mul_g(const uint8_t k_be[32], uint8_t out[64]) {
    ensure_init();
    uint64_t k[4];
    be_to_limbs(k_be, k);
    jac_t r;
    aff_t a;
    jac_mul(&r, &GEN_ONE, k);
    jac_to_aff(&a, &r);
    fr_to_be_bytes(&a.x, out);
    fr_to_be_bytes(&a.y, out + 32);
    return a.inf;
}
void oracle_pedersen_compress(const uint8_t *inputs_be32, uint32_t m, uint8_t out[32]) {
    ensure_init();
    fr_t v[8], c;
    for (uint32_t i = 0; i < m && i < 8; i++) fr_from_be_bytes_reduce(&v[i], inputs_be32 + 32 * i, 32);
    compress(&c, v, (int)(m < 8 ? m : 8));
    fr_to_be_bytes(&c, out);
}
void oracle_pedersen_hash_single(const uint8_t v_be[32], int parity, uint8_t out[64]) {
    ensure_init();
    fr_t v;
    jac_t j;
    aff_t a;
    fr_from_be_bytes_reduce(&v, v_be, 32);
    hash_single(&j, &v, parity);
    jac_to_aff(&a, &j);
    fr_to_be_bytes(&a.x, out);
    fr_to_be_bytes(&a.y, out + 32);
}
int oracle_pedersen(const uint8_t *inputs_be32, uint32_t n, uint32_t hash_index, uint8_t out[64]) {
    fr_t in[64], x, y;
    char err[8];
    if (n > 64) return -1;
    for (uint32_t i = 0; i < n; i++) fr_from_be_bytes_reduce(&in[i], inputs_be32 + 32 * i, 32);
    bb_pedersen(0, in, n, hash_index, &x, &y, err, sizeof err);
    fr_to_be_bytes(&x, out);
    fr_to_be_bytes(&y, out + 32);
    return 0;
}
int oracle_fixed_base(const uint8_t low_be[32], const uint8_t high_be[32], uint8_t out[64], char *err, size_t errlen) {
    fr_t lo, hi, x, y;
    fr_from_be_bytes_reduce(&lo, low_be, 32);
    fr_from_be_bytes_reduce(&hi, high_be, 32);
    int rc = bb_fixed_base(0, &lo, &hi, &x, &y, err, errlen);
    if (rc) return rc;
    fr_to_be_bytes(&x, out);
    fr_to_be_bytes(&y, out + 32);
    return 0;
}
int oracle_schnorr_verify(const uint8_t pk_be[64], const uint8_t *sig, size_t sig_len, const uint8_t *msg, size_t msg_len) {
    fr_t x, y;
    char err[128];
    int ok = 0;
    fr_from_be_bytes_reduce(&x, pk_be, 32);
    fr_from_be_bytes_reduce(&y, pk_be + 32, 32);
    int rc = bb_schnorr(0, &x, &y, sig, sig_len, msg, msg_len, &ok, err, sizeof err);
    return rc ? -rc : ok;
}
/* matching signer for synthetic inputs: pk = sk*G, R = k*G, e = blake2s(be32(compress(R.x, pk.x, pk.y)) || msg),
 * s = (k - sk * (e mod q)) mod q. sk, k: 32-byte big-endian integers < q. out: pk (64) || sig (64). */
int oracle_schnorr_sign(const uint8_t sk_be[32], const uint8_t k_be[32], const uint8_t *msg, size_t msg_len, uint8_t out[128]) {
    ensure_init();
    if (msg_len > 800) return -1;
    uint64_t sk[4], k[4], e[4];
    be_to_limbs(sk_be, sk);
    be_to_limbs(k_be, k);
    reduce_mod_q(sk);
    reduce_mod_q(k);
    jac_t pj, rj;
    aff_t pk, R;
    jac_mul(&pj, &GEN_ONE, sk);
    jac_to_aff(&pk, &pj);
    jac_mul(&rj, &GEN_ONE, k);
    jac_to_aff(&R, &rj);
    if (pk.inf || R.inf) return -1;
    uint8_t eb[32];
    challenge(eb, &R.x, &pk.x, &pk.y, msg, msg_len);
    be_to_limbs(eb, e);
    reduce_mod_q(e);
    /* prod = sk * e mod q by double-and-add on integers mod q */
    uint64_t prod[4] = {0, 0, 0, 0};
    for (int i = 255; i >= 0; i--) {
        /* prod = 2 * prod mod q */
        uint64_t c = 0;
        for (int j = 0; j < 4; j++) { uint64_t n = (prod[j] << 1) | c; c = prod[j] >> 63; prod[j] = n; }
        reduce_mod_q(prod); /* prod < 2q < 2^256: carry c is 0 since q < 2^254 */
        if ((e[i / 64] >> (i % 64)) & 1) {
            unsigned __int128 cc = 0;
            for (int j = 0; j < 4; j++) { cc += (unsigned __int128)prod[j] + sk[j]; prod[j] = (uint64_t)cc; cc >>= 64; }
            reduce_mod_q(prod);
        }
    }
    /* s = k - prod mod q */
    uint64_t s[4], borrow = 0;
    for (int j = 0; j < 4; j++) {
        unsigned __int128 d = (unsigned __int128)k[j] - prod[j] - borrow;
        s[j] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
    if (borrow) {
        unsigned __int128 cc = 0;
        for (int j = 0; j < 4; j++) { cc += (unsigned __int128)s[j] + GQ[j]; s[j] = (uint64_t)cc; cc >>= 64; }
    }
    fr_to_be_bytes(&pk.x, out);
    fr_to_be_bytes(&pk.y, out + 32);
    for (int i = 0; i < 32; i++) out[64 + 31 - i] = (uint8_t)(s[i / 8] >> (8 * (i % 8)));
    memcpy(out + 96, eb, 32);
    return 0;
}
