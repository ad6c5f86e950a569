// secp_device.hpp -- arithmetic of secp256k1 / secp256r1 for the ECDSA opcodes (ops_ecdsa.hpp), one lane per verification.
//   blackbox_solver/src/lib.rs:66-210 (verify_secp256k1/r1_ecdsa_signature; k256 0.11.6 / p256 0.11.1 = SEC 1 v2 section 4.1.4).
// Everything is __host__ __device__ with compile-time moduli, so tools/secp_device_host_test.hip runs the same code on the host against
// Python integers (tests/test_secp_device_on_host.py).
//   * base field: plain residues in [0, p), 8 x 32-bit limbs. The product is a 64-multiply schoolbook (squares: 36) followed by the
//     reduction the SHAPE of the prime allows -- p = 2^256 - 2^32 - 977 folds the high half in with one 8-limb multiply by 977 and a
//     shift; p = 2^256 - 2^224 + 2^192 + 2^96 - 1 (NIST P-256) with nine signed word sums (FIPS 186-4 D.2.3) -- instead of the 64 further
//     multiplies of a Montgomery reduction.
//   * scalar field (three products per verification): Montgomery, modulus a compile-time constant.
//   * inversions: safegcd (fr_device.hpp fr_safegcd_inv) with the modulus as a template parameter.
//   * square root for the decompression of the public key (both p = 3 mod 4): addition chains for (p + 1) / 4 (253 squarings + 13 / 7 products).
//   * u1 G + u2 Q: u2 Q on signed 4-bit windows over a per-lane table {Q .. 8Q} normalised to affine with ONE inversion (256 doublings +
//     65 mixed additions, every lane adds at the same steps); u1 G as 16 mixed additions from a precomputed table d * 2^(16 j) * G
//     (16 x 65 535 affine points per curve, 64 MiB, built once per device by secp_gtable_entry) onto the same accumulator.
#pragma once
#include "fr_device.hpp"

namespace acvm {

// ---- constants (C: 0 = secp256k1, 1 = secp256r1)
template <int C>
struct Secp {
    static FR_HD __forceinline__ uint32_t p(int i) {
        constexpr uint32_t L[2][8] = {{0xfffffc2fu, 0xfffffffeu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu},
                                      {0xffffffffu, 0xffffffffu, 0xffffffffu, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000001u, 0xffffffffu}};
        return L[C][i];
    }
    static FR_HD __forceinline__ uint32_t n(int i) {
        constexpr uint32_t L[2][8] = {{0xd0364141u, 0xbfd25e8cu, 0xaf48a03bu, 0xbaaedce6u, 0xfffffffeu, 0xffffffffu, 0xffffffffu, 0xffffffffu},
                                      {0xfc632551u, 0xf3b9cac2u, 0xa7179e84u, 0xbce6faadu, 0xffffffffu, 0xffffffffu, 0x00000000u, 0xffffffffu}};
        return L[C][i];
    }
    static FR_HD __forceinline__ uint32_t half_n(int i) {  // floor(n / 2)
        constexpr uint32_t L[2][8] = {{0x681b20a0u, 0xdfe92f46u, 0x57a4501du, 0x5d576e73u, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x7fffffffu},
                                      {0x7e3192a8u, 0x79dce561u, 0xd38bcf42u, 0xde737d56u, 0xffffffffu, 0x7fffffffu, 0x80000000u, 0x7fffffffu}};
        return L[C][i];
    }
    static FR_HD __forceinline__ uint32_t r2n(int i) {  // 2^512 mod n
        constexpr uint32_t L[2][8] = {{0x67d7d140u, 0x896cf214u, 0x0e7cf878u, 0x741496c2u, 0x5bcd07c6u, 0xe697f5e4u, 0x81c69bc5u, 0x9d671cd5u},
                                      {0xbe79eea2u, 0x83244c95u, 0x49bd6fa6u, 0x4699799cu, 0x2b6bec59u, 0x2845b239u, 0xf3d95620u, 0x66e12d94u}};
        return L[C][i];
    }
    static constexpr uint32_t NINV = C == 0 ? 0x5588b13fu : 0xee00bc4fu;  // -n^-1 mod 2^32
    static FR_HD __forceinline__ uint32_t b(int i) {
        constexpr uint32_t L[2][8] = {{7u, 0u, 0u, 0u, 0u, 0u, 0u, 0u},
                                      {0x27d2604bu, 0x3bce3c3eu, 0xcc53b0f6u, 0x651d06b0u, 0x769886bcu, 0xb3ebbd55u, 0xaa3a93e7u, 0x5ac635d8u}};
        return L[C][i];
    }
    static FR_HD __forceinline__ uint32_t gx(int i) {
        constexpr uint32_t L[2][8] = {{0x16f81798u, 0x59f2815bu, 0x2dce28d9u, 0x029bfcdbu, 0xce870b07u, 0x55a06295u, 0xf9dcbbacu, 0x79be667eu},
                                      {0xd898c296u, 0xf4a13945u, 0x2deb33a0u, 0x77037d81u, 0x63a440f2u, 0xf8bce6e5u, 0xe12c4247u, 0x6b17d1f2u}};
        return L[C][i];
    }
    static FR_HD __forceinline__ uint32_t gy(int i) {
        constexpr uint32_t L[2][8] = {{0xfb10d4b8u, 0x9c47d08fu, 0xa6855419u, 0xfd17b448u, 0x0e1108a8u, 0x5da4fbfcu, 0x26a3c465u, 0x483ada77u},
                                      {0x37bf51f5u, 0xcbb64068u, 0x6b315eceu, 0x2bce3357u, 0x7c0f9e16u, 0x8ee7eb4au, 0xfe1a7f9bu, 0x4fe342e2u}};
        return L[C][i];
    }
};
// the 9 x 30-bit signed-limb form of the four moduli for safegcd; index 2 * curve + (0: p, 1: n)
template <int K>
struct SecpMod30 {
    static FR_HD __forceinline__ int32_t p30(int i) {
        constexpr int32_t L[4][9] = {
            {0x3ffffc2f, 0x3ffffffb, 0x3fffffff, 0x3fffffff, 0x3fffffff, 0x3fffffff, 0x3fffffff, 0x3fffffff, 0xffff},
            {0x10364141, 0x3f497a33, 0x348a03bb, 0x2bb739ab, 0x3ffffeba, 0x3fffffff, 0x3fffffff, 0x3fffffff, 0xffff},
            {0x3fffffff, 0x3fffffff, 0x3fffffff, 0x0000003f, 0x00000000, 0x00000000, 0x00001000, 0x3fffc000, 0xffff},
            {0x3c632551, 0x0ee72b0b, 0x3179e84f, 0x39beab69, 0x3fffffbc, 0x3fffffff, 0x00000fff, 0x3fffc000, 0xffff}};
        return L[K][i];
    }
    static constexpr uint32_t PINV30 = K == 0 ? 0x2ddacacfu : K == 1 ? 0x2a774ec1u : K == 2 ? 0x3fffffffu : 0x11ff43b1u;
};

template <class F>
FR_HD __forceinline__ Fr secp_limbs(F f) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = f(i);
    return r;
}
template <int C> FR_HD __forceinline__ Fr sp_modulus() { return secp_limbs([](int i) { return Secp<C>::p(i); }); }
template <int C> FR_HD __forceinline__ Fr sn_modulus() { return secp_limbs([](int i) { return Secp<C>::n(i); }); }
FR_HD __forceinline__ bool secp_geq(const Fr &a, const Fr &b) {
    Fr d;
    return fr_sub256(d, a, b) == 0;
}

// ---- wide products, column by column: a column's 64-bit products are summed in a 96-bit accumulator (acc, top). On the device one term is
// v_mad_u64_u32 (the 64-bit accumulate of the multiplier, carry out in vcc) + v_addc_co_u32; written in C the compiler builds the 64-bit
// addend of every term from a zeroed register pair (132 v_mov + 49 64-bit adds beside the 64 multiplies of one product: 245 instructions;
// this form: 164).
FR_HD __forceinline__ void secp_mac(uint64_t &acc, uint32_t &top, uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(top) : "v"(a), "v"(b) : "vcc");
#else
    const uint64_t p = (uint64_t)a * b, s = acc + p;
    top += s < p ? 1u : 0u;
    acc = s;
#endif
}
FR_HD __forceinline__ void secp_mul_wide(uint32_t t[16], const Fr &a, const Fr &b) {
    uint64_t acc = 0;
    uint32_t top = 0;
#pragma unroll
    for (int k = 0; k < 15; k++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int j = k - i;
            if (j < 0 || j > 7) continue;
            secp_mac(acc, top, a.v[i], b.v[j]);
        }
        t[k] = (uint32_t)acc;
        acc = acc >> 32 | (uint64_t)top << 32;
        top = 0;
    }
    t[15] = (uint32_t)acc;
}
// 28 cross products, doubled, + 8 squares
FR_HD __forceinline__ void secp_sqr_wide(uint32_t t[16], const Fr &a) {
    uint64_t acc = 0;
    uint32_t top = 0;
    t[0] = 0;
#pragma unroll
    for (int k = 1; k < 14; k++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int j = k - i;
            if (j <= i || j > 7) continue;
            secp_mac(acc, top, a.v[i], a.v[j]);
        }
        t[k] = (uint32_t)acc;
        acc = acc >> 32 | (uint64_t)top << 32;
        top = 0;
    }
    t[14] = (uint32_t)acc;
    t[15] = (uint32_t)(acc >> 32);
#pragma unroll
    for (int i = 15; i > 0; i--) t[i] = t[i] << 1 | t[i - 1] >> 31;
    t[0] = 0;
    uint32_t cin = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {  // + a_i^2 at words 2i, 2i + 1, the carry of a pair into the next
        uint64_t pair = (uint64_t)t[2 * i] | (uint64_t)t[2 * i + 1] << 32;
        uint32_t over = 0;
        secp_mac(pair, over, a.v[i], a.v[i]);
        const uint64_t s = pair + cin;
        over += s < pair ? 1u : 0u;
        t[2 * i] = (uint32_t)s;
        t[2 * i + 1] = (uint32_t)(s >> 32);
        cin = over;
    }
}

// ---- reduction of a 512-bit product to [0, p)
template <int C>
FR_HD __forceinline__ Fr sp_reduce(const uint32_t t[16]);
// secp256k1: 2^256 = 2^32 + 977 (mod p)
template <>
FR_HD __forceinline__ Fr sp_reduce<0>(const uint32_t t[16]) {
    uint32_t r[8];
    uint64_t acc = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {  // low + high * 977 + (high << 32)
        acc += (uint64_t)t[8 + i] * 977u + t[i];
        if (i > 0) acc += t[7 + i];
        r[i] = (uint32_t)acc;
        acc >>= 32;
    }
    acc += t[15];  // what is left above 2^256: < 2^34
    const uint64_t k = acc;
    acc = k * 977u + r[0];
    r[0] = (uint32_t)acc;
    acc >>= 32;
    acc += (uint64_t)r[1] + (uint32_t)k;
    r[1] = (uint32_t)acc;
    acc >>= 32;
    acc += (uint64_t)r[2] + (k >> 32);
    r[2] = (uint32_t)acc;
    acc >>= 32;
#pragma unroll
    for (int i = 3; i < 8; i++) {
        acc += r[i];
        r[i] = (uint32_t)acc;
        acc >>= 32;
    }
    const uint32_t over = (uint32_t)acc;  // 0 / 1; the wrapped value is then tiny, so r + (2^32 + 977) neither carries nor reaches p
    Fr x, d, c977 = fr_zero();
#pragma unroll
    for (int i = 0; i < 8; i++) x.v[i] = r[i];
    c977.v[0] = 977u;
    c977.v[1] = 1u;
    const uint32_t carry = fr_add256(d, x, c977);  // x >= p <=> x + (2^256 - p) carries
    const bool take = (over | carry) != 0u;
#pragma unroll
    for (int i = 0; i < 8; i++) x.v[i] = take ? d.v[i] : x.v[i];
    return x;
}
// secp256r1: FIPS 186-4 D.2.3 on 32-bit words c0..c15, as signed column sums; 2^256 = 2^224 - 2^192 - 2^96 + 1 (mod p) folds the carry
template <>
FR_HD __forceinline__ Fr sp_reduce<1>(const uint32_t t[16]) {
    int64_t c[16];
#pragma unroll
    for (int i = 0; i < 16; i++) c[i] = (int64_t)t[i];
    int64_t w[8];
    w[0] = c[0] + c[8] + c[9] - c[11] - c[12] - c[13] - c[14];
    w[1] = c[1] + c[9] + c[10] - c[12] - c[13] - c[14] - c[15];
    w[2] = c[2] + c[10] + c[11] - c[13] - c[14] - c[15];
    w[3] = c[3] + 2 * c[11] + 2 * c[12] + c[13] - c[15] - c[8] - c[9];
    w[4] = c[4] + 2 * c[12] + 2 * c[13] + c[14] - c[9] - c[10];
    w[5] = c[5] + 2 * c[13] + 2 * c[14] + c[15] - c[10] - c[11];
    w[6] = c[6] + 3 * c[14] + 2 * c[15] + c[13] - c[8] - c[9];
    w[7] = c[7] + 3 * c[15] + c[8] - c[10] - c[11] - c[12] - c[13];
    uint32_t r[8];
    int64_t acc = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        acc += w[i];
        r[i] = (uint32_t)acc;
        acc >>= 32;
    }
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {  // |carry| <= 6, then <= 1, then 0
        const int64_t k = acc;
        acc = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            acc += (int64_t)r[i];
            if (i == 0 || i == 7) acc += k;
            if (i == 3 || i == 6) acc -= k;
            r[i] = (uint32_t)acc;
            acc >>= 32;
        }
    }
    Fr x, d;
#pragma unroll
    for (int i = 0; i < 8; i++) x.v[i] = r[i];
    const uint32_t borrow = fr_sub256(d, x, sp_modulus<1>());
#pragma unroll
    for (int i = 0; i < 8; i++) x.v[i] = borrow ? x.v[i] : d.v[i];
    return x;
}

template <int C>
FR_HD __forceinline__ Fr sp_mul(const Fr &a, const Fr &b) {
    uint32_t t[16];
    secp_mul_wide(t, a, b);
    return sp_reduce<C>(t);
}
template <int C>
FR_HD __forceinline__ Fr sp_sqr(const Fr &a) {
    uint32_t t[16];
    secp_sqr_wide(t, a);
    return sp_reduce<C>(t);
}
template <int C>
FR_HD __forceinline__ Fr sp_add(const Fr &a, const Fr &b) {
    Fr r, d;
    const uint32_t c = fr_add256(r, a, b);
    const uint32_t borrow = fr_sub256(d, r, sp_modulus<C>());
    const bool sub = c != 0u || borrow == 0u;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = sub ? d.v[i] : r.v[i];
    return r;
}
template <int C>
FR_HD __forceinline__ Fr sp_sub(const Fr &a, const Fr &b) {
    Fr r, q;
    const uint32_t mask = fr_sub256(r, a, b) ? 0xffffffffu : 0u;
#pragma unroll
    for (int i = 0; i < 8; i++) q.v[i] = Secp<C>::p(i) & mask;
    fr_add256(r, r, q);
    return r;
}
template <int C>
FR_HD __forceinline__ Fr sp_neg(const Fr &a) { return sp_sub<C>(fr_zero(), a); }
template <int C>
FR_HD __forceinline__ Fr sp_sqr_n(Fr a, int n) {
#pragma unroll 1
    for (int i = 0; i < n; i++) a = sp_sqr<C>(a);
    return a;
}
template <int C>
FR_HD inline __noinline__ Fr sp_inv(const Fr &a) { return fr_safegcd_inv<SecpMod30<2 * C>>(a); }  // 0 for 0
template <int C>
FR_HD inline __noinline__ Fr sn_inv(const Fr &a) { return fr_safegcd_inv<SecpMod30<2 * C + 1>>(a); }

// a^((p + 1) / 4): the square root of a when a is a square (both primes are 3 mod 4)
template <int C>
FR_HD inline __noinline__ Fr sp_sqrt_candidate(const Fr &a);
// (p + 1) / 4 = 1{223} 0 1{22} 0000 11 00 in binary (the chain of libsecp256k1's field square root)
template <>
FR_HD inline __noinline__ Fr sp_sqrt_candidate<0>(const Fr &a) {
    const Fr x2 = sp_mul<0>(sp_sqr<0>(a), a), x3 = sp_mul<0>(sp_sqr<0>(x2), a);
    const Fr x6 = sp_mul<0>(sp_sqr_n<0>(x3, 3), x3), x9 = sp_mul<0>(sp_sqr_n<0>(x6, 3), x3), x11 = sp_mul<0>(sp_sqr_n<0>(x9, 2), x2);
    const Fr x22 = sp_mul<0>(sp_sqr_n<0>(x11, 11), x11), x44 = sp_mul<0>(sp_sqr_n<0>(x22, 22), x22), x88 = sp_mul<0>(sp_sqr_n<0>(x44, 44), x44);
    const Fr x176 = sp_mul<0>(sp_sqr_n<0>(x88, 88), x88), x220 = sp_mul<0>(sp_sqr_n<0>(x176, 44), x44), x223 = sp_mul<0>(sp_sqr_n<0>(x220, 3), x3);
    Fr t = sp_mul<0>(sp_sqr_n<0>(x223, 23), x22);
    t = sp_mul<0>(sp_sqr_n<0>(t, 6), x2);
    return sp_sqr_n<0>(t, 2);
}
// (p + 1) / 4 = (2^32 - 1) 2^222 + 2^190 + 2^94
template <>
FR_HD inline __noinline__ Fr sp_sqrt_candidate<1>(const Fr &a) {
    const Fr x2 = sp_mul<1>(sp_sqr<1>(a), a), x4 = sp_mul<1>(sp_sqr_n<1>(x2, 2), x2), x8 = sp_mul<1>(sp_sqr_n<1>(x4, 4), x4);
    const Fr x16 = sp_mul<1>(sp_sqr_n<1>(x8, 8), x8), x32 = sp_mul<1>(sp_sqr_n<1>(x16, 16), x16);
    Fr t = sp_mul<1>(sp_sqr_n<1>(x32, 32), a);
    t = sp_mul<1>(sp_sqr_n<1>(t, 96), a);
    return sp_sqr_n<1>(t, 94);
}

// ---- scalar field: Montgomery product a b / 2^256 mod n (CIOS with the extra carry word: n is a full 256-bit modulus)
template <int C>
FR_HD inline __noinline__ Fr sn_mont(const Fr &a, const Fr &b) {
    uint32_t t[10];
#pragma unroll
    for (int i = 0; i < 10; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            c += (uint64_t)a.v[j] * b.v[i] + t[j];
            t[j] = (uint32_t)c;
            c >>= 32;
        }
        c += t[8];
        t[8] = (uint32_t)c;
        t[9] = (uint32_t)(c >> 32);
        const uint32_t m = t[0] * Secp<C>::NINV;
        c = ((uint64_t)m * Secp<C>::n(0) + t[0]) >> 32;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            c += (uint64_t)m * Secp<C>::n(j) + t[j];
            t[j - 1] = (uint32_t)c;
            c >>= 32;
        }
        c += t[8];
        t[7] = (uint32_t)c;
        t[8] = t[9] + (uint32_t)(c >> 32);
    }
    Fr r, d;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = t[i];
    const uint32_t borrow = fr_sub256(d, r, sn_modulus<C>());
    const bool sub = t[8] != 0u || borrow == 0u;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = sub ? d.v[i] : r.v[i];
    return r;
}
// a b mod n for plain residues
template <int C>
FR_HD __forceinline__ Fr sn_mul(const Fr &a, const Fr &b) {
    return sn_mont<C>(sn_mont<C>(a, b), secp_limbs([](int i) { return Secp<C>::r2n(i); }));
}

template <int C>
FR_HD __forceinline__ Fr sn_add(const Fr &a, const Fr &b) {
    Fr r, d;
    const uint32_t c = fr_add256(r, a, b);
    const uint32_t borrow = fr_sub256(d, r, sn_modulus<C>());
    const bool sub = c != 0u || borrow == 0u;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = sub ? d.v[i] : r.v[i];
    return r;
}
template <int C>
FR_HD __forceinline__ Fr sn_sub(const Fr &a, const Fr &b) {
    Fr r, q;
    const uint32_t mask = fr_sub256(r, a, b) ? 0xffffffffu : 0u;
#pragma unroll
    for (int i = 0; i < 8; i++) q.v[i] = Secp<C>::n(i) & mask;
    fr_add256(r, r, q);
    return r;
}

// ---- secp256k1 only: the endomorphism (x, y) -> (beta x, y) = lambda (x, y). k = k1 + k2 lambda (mod n) with |k1|, |k2| < 2^128 halves the
// doublings of a variable-base product. The split is the one of libsecp256k1 (scalar_split_lambda): with the lattice basis (a1, b1), (a2, b2)
// of {(x, y): x + y lambda = 0 mod n}, c1 = round(k g1 / 2^384), c2 = round(k g2 / 2^384) for g1 = round(2^384 b2 / n), g2 = round(2^384 (-b1) / n),
// k2 = c1 (-b1) + c2 (-b2), k1 = k - k2 lambda. Returned as magnitudes < 2^128 and signs. (Constants re-derived and the bound checked over
// 200 000 scalars by the script quoted in DESIGN.md section 9; tests/test_secp_device_on_host.py checks k1 + k2 lambda = k.)
struct SecpSplit { Fr k1, k2; bool neg1, neg2; };
FR_HD __forceinline__ Fr secp_round_shift384(const uint32_t t[16]) {  // (t + 2^383) >> 384 of a 512-bit product
    Fr c = fr_zero();
    uint64_t acc = (uint64_t)t[12] + (t[11] >> 31);
    c.v[0] = (uint32_t)acc;
    acc >>= 32;
#pragma unroll
    for (int i = 1; i < 4; i++) {
        acc += t[12 + i];
        c.v[i] = (uint32_t)acc;
        acc >>= 32;
    }
    c.v[4] = (uint32_t)acc;
    return c;
}
FR_HD inline __noinline__ SecpSplit secp256k1_split_lambda(const Fr &k) {
    const Fr g1 = {{0x45dbb031u, 0xe893209au, 0x71e8ca7fu, 0x3daa8a14u, 0x9284eb15u, 0xe86c90e4u, 0xa7d46bcdu, 0x3086d221u}};
    const Fr g2 = {{0x8ac47f71u, 0x1571b4aeu, 0x9df506c6u, 0x221208acu, 0x0abfe4c4u, 0x6f547fa9u, 0x010e8828u, 0xe4437ed6u}};
    const Fr mb1 = {{0x0abfe4c3u, 0x6f547fa9u, 0x010e8828u, 0xe4437ed6u, 0u, 0u, 0u, 0u}};
    const Fr mb2 = {{0x3db1562cu, 0xd765cda8u, 0x0774346du, 0x8a280ac5u, 0xfffffffeu, 0xffffffffu, 0xffffffffu, 0xffffffffu}};
    const Fr lam = {{0x1b23bd72u, 0xdf02967cu, 0x20816678u, 0x122e22eau, 0x8812645au, 0xa5261c02u, 0xc05c30e0u, 0x5363ad4cu}};
    uint32_t t[16];
    secp_mul_wide(t, k, g1);
    const Fr c1 = secp_round_shift384(t);
    secp_mul_wide(t, k, g2);
    const Fr c2 = secp_round_shift384(t);
    SecpSplit r;
    r.k2 = sn_add<0>(sn_mul<0>(c1, mb1), sn_mul<0>(c2, mb2));
    r.k1 = sn_sub<0>(k, sn_mul<0>(r.k2, lam));
    const Fr half = secp_limbs([](int i) { return Secp<0>::half_n(i); });
    Fr d;
    r.neg1 = fr_sub256(d, half, r.k1) != 0u;  // k1 > n / 2: the small representative is n - k1
    if (r.neg1) r.k1 = sn_sub<0>(fr_zero(), r.k1);
    r.neg2 = fr_sub256(d, half, r.k2) != 0u;
    if (r.neg2) r.k2 = sn_sub<0>(fr_zero(), r.k2);
    return r;
}

// ---- points: Jacobian (X / Z^2, Y / Z^3), Z == 0 <=> identity; a = 0 (secp256k1) or a = -3 (secp256r1)
struct SJac { Fr X, Y, Z; };
struct SAff { Fr x, y; };
FR_HD __forceinline__ Fr secp_one() {
    Fr o = fr_zero();
    o.v[0] = 1u;
    return o;
}
FR_HD __forceinline__ SJac sj_identity() { return SJac{secp_one(), secp_one(), fr_zero()}; }
template <int C>
FR_HD __forceinline__ SJac sj_dbl(const SJac &p) {
    if (fr_is_zero(p.Z) || fr_is_zero(p.Y)) return sj_identity();
    const Fr yy = sp_sqr<C>(p.Y), yyyy = sp_sqr<C>(yy);
    Fr s = sp_mul<C>(p.X, yy);
    s = sp_add<C>(s, s);
    s = sp_add<C>(s, s);
    Fr m;
    if (C == 1) {  // 3 X^2 - 3 Z^4 = 3 (X - Z^2)(X + Z^2)
        const Fr zz = sp_sqr<C>(p.Z);
        m = sp_mul<C>(sp_sub<C>(p.X, zz), sp_add<C>(p.X, zz));
    } else m = sp_sqr<C>(p.X);
    m = sp_add<C>(sp_add<C>(m, m), m);
    SJac r;
    r.X = sp_sub<C>(sp_sub<C>(sp_sqr<C>(m), s), s);
    Fr y8 = sp_add<C>(yyyy, yyyy);
    y8 = sp_add<C>(y8, y8);
    y8 = sp_add<C>(y8, y8);
    r.Y = sp_sub<C>(sp_mul<C>(m, sp_sub<C>(s, r.X)), y8);
    const Fr yz = sp_mul<C>(p.Y, p.Z);
    r.Z = sp_add<C>(yz, yz);
    return r;
}
// complete mixed addition of a finite affine point: 8 M + 3 S
template <int C>
FR_HD __forceinline__ SJac sj_add_aff(const SJac &p, const SAff &q) {
    if (fr_is_zero(p.Z)) return SJac{q.x, q.y, secp_one()};
    const Fr z1z1 = sp_sqr<C>(p.Z);
    const Fr u2 = sp_mul<C>(q.x, z1z1), s2 = sp_mul<C>(sp_mul<C>(q.y, p.Z), z1z1);
    const Fr h = sp_sub<C>(u2, p.X), rr = sp_sub<C>(s2, p.Y);
    if (fr_is_zero(h)) return fr_is_zero(rr) ? sj_dbl<C>(p) : sj_identity();
    const Fr hh = sp_sqr<C>(h), hhh = sp_mul<C>(hh, h), v = sp_mul<C>(p.X, hh);
    SJac r;
    r.X = sp_sub<C>(sp_sub<C>(sp_sub<C>(sp_sqr<C>(rr), hhh), v), v);
    r.Y = sp_sub<C>(sp_mul<C>(rr, sp_sub<C>(v, r.X)), sp_mul<C>(p.Y, hhh));
    r.Z = sp_mul<C>(p.Z, h);
    return r;
}
template <int C>
FR_HD __forceinline__ SAff sj_to_affine(const SJac &p) {  // p finite
    const Fr zi = sp_inv<C>(p.Z), zi2 = sp_sqr<C>(zi);
    return SAff{sp_mul<C>(p.X, zi2), sp_mul<C>(p.Y, sp_mul<C>(zi2, zi))};
}
template <int C>
FR_HD __forceinline__ SAff secp_generator() {
    return SAff{secp_limbs([](int i) { return Secp<C>::gx(i); }), secp_limbs([](int i) { return Secp<C>::gy(i); })};
}

// ---- the table of the generator: entry (j, d) = d * 2^(W j) * G, j < 256 / W, 1 <= d < 2^W, 16 words (x, y) at (((j << W) + d) * 16).
// W = SECP_GWIN_BITS: 16 on the device (16 additions for u1 G, 64 MiB per curve); the host-run test builds its table with -DSECP_GWIN_BITS=8.
#ifndef SECP_GWIN_BITS
#define SECP_GWIN_BITS 16
#endif
constexpr uint32_t SECP_GWIN = SECP_GWIN_BITS, SECP_GWINDOWS = 256u / SECP_GWIN;
static_assert(SECP_GWIN == 8u || SECP_GWIN == 16u, "window of the generator table");
constexpr uint32_t SECP_GTABLE_WORDS = (SECP_GWINDOWS << SECP_GWIN) * 16u;  // per curve
template <int C>
FR_HD inline SAff secp_gtable_entry(uint32_t j, uint32_t d) {
    const SAff G = secp_generator<C>();
    SJac base{G.x, G.y, secp_one()};
    for (uint32_t i = 0; i < SECP_GWIN * j; i++) base = sj_dbl<C>(base);
    const SAff B = sj_to_affine<C>(base);
    SJac acc = sj_identity();
    for (int i = (int)SECP_GWIN - 1; i >= 0; i--) {
        acc = sj_dbl<C>(acc);
        if ((d >> i) & 1u) acc = sj_add_aff<C>(acc, B);
    }
    return sj_to_affine<C>(acc);
}

// ---- u1 G + u2 Q
// tab <- {Q, 2Q, .., 8Q} affine: seven Jacobian multiples, their Z inverted together (Montgomery's trick: 3 products per point + one inversion)
template <int C>
FR_HD __forceinline__ void secp_window_table(SAff tab[8], const SAff &Q) {
    Fr zs[8], pre[8];
    tab[0] = Q;
    SJac cur = sj_dbl<C>(SJac{Q.x, Q.y, secp_one()});
    tab[1] = SAff{cur.X, cur.Y};
    zs[1] = pre[1] = cur.Z;
#pragma unroll 1
    for (int k = 2; k < 8; k++) {
        cur = sj_add_aff<C>(cur, Q);
        tab[k] = SAff{cur.X, cur.Y};
        zs[k] = cur.Z;
        pre[k] = sp_mul<C>(pre[k - 1], cur.Z);
    }
    // k Q is finite for 1 <= k <= 8 (prime order > 8), so every Z is invertible
    Fr inv = sp_inv<C>(pre[7]);
#pragma unroll 1
    for (int k = 7; k >= 1; k--) {
        const Fr zi = k > 1 ? sp_mul<C>(inv, pre[k - 1]) : inv;
        if (k > 1) inv = sp_mul<C>(inv, zs[k]);
        const Fr zi2 = sp_sqr<C>(zi);
        tab[k] = SAff{sp_mul<C>(tab[k].x, zi2), sp_mul<C>(tab[k].y, sp_mul<C>(zi2, zi))};
    }
}
FR_HD __forceinline__ uint32_t secp_limb_at(const Fr &a, uint32_t k) {  // a.v[k] for a lane-dependent k without indexing the registers
    uint32_t w = 0;
#pragma unroll
    for (uint32_t i = 0; i < 8; i++) w = i == k ? a.v[i] : w;
    return w;
}
// acc + u1 G from the table of the generator
template <int C>
FR_HD __forceinline__ SJac secp_add_generator_multiple(SJac acc, const Fr &u1, const uint32_t *__restrict__ gtab) {
#pragma unroll 1
    for (uint32_t j = 0; j < SECP_GWINDOWS; j++) {
        const uint32_t d = (secp_limb_at(u1, (SECP_GWIN * j) >> 5) >> ((SECP_GWIN * j) & 31u)) & ((1u << SECP_GWIN) - 1u);
        if (d != 0u) {
            const uint32_t *g = gtab + (size_t)((j << SECP_GWIN) + d) * 16u;
            SAff q;
#pragma unroll
            for (int k = 0; k < 8; k++) { q.x.v[k] = g[k]; q.y.v[k] = g[8 + k]; }
            acc = sj_add_aff<C>(acc, q);
        }
    }
    return acc;
}
template <int C>
FR_HD inline __noinline__ SJac secp_mul2(const Fr &u1, const SAff &Q, const Fr &u2, const uint32_t *__restrict__ gtab) {
    SAff tab[8];
    secp_window_table<C>(tab, Q);
    Fr eights;
#pragma unroll
    for (int i = 0; i < 8; i++) eights.v[i] = 0x88888888u;
    SJac acc = sj_identity();
    if constexpr (C == 0) {
        // u2 = k1 + k2 lambda: two 128-bit ladders on one accumulator, 128 doublings; lambda (k Q) = (beta x, y) of the same table
        const Fr beta = {{0x719501eeu, 0xc1396c28u, 0x12f58995u, 0x9cf04975u, 0xac3434e9u, 0x6e64479eu, 0x657c0710u, 0x7ae96a2bu}};
        Fr bx[8];
#pragma unroll 1
        for (int k = 0; k < 8; k++) bx[k] = sp_mul<C>(tab[k].x, beta);
        const SecpSplit sp = secp256k1_split_lambda(u2);
        // signed digits: e = k + 0x88..8 over 32 nibbles, digit i = nibble i of e - 8 (i < 32), digit 32 = bit 128 of e
        Fr e1, e2, eights128 = eights;
#pragma unroll
        for (int i = 4; i < 8; i++) eights128.v[i] = 0u;
        fr_add256(e1, sp.k1, eights128);
        fr_add256(e2, sp.k2, eights128);
#pragma unroll 1
        for (int i = 128; i >= 0; i--) {
            if (i != 128) acc = sj_dbl<C>(acc);
            if (i & 3) continue;
#pragma unroll 1
            for (int h = 0; h < 2; h++) {
                Fr e;
#pragma unroll
                for (int k = 0; k < 8; k++) e.v[k] = h ? e2.v[k] : e1.v[k];
                const uint32_t nib = (secp_limb_at(e, (uint32_t)i >> 5) >> (i & 31)) & 15u;
                int32_t dg = i == 128 ? (int32_t)(nib & 1u) : (int32_t)nib - 8;
                if (h ? sp.neg2 : sp.neg1) dg = -dg;
                if (dg != 0) {
                    const uint32_t mag = (uint32_t)(dg < 0 ? -dg : dg);
                    SAff q = tab[mag - 1u];
                    if (h) q.x = bx[mag - 1u];
                    if (dg < 0) q.y = sp_neg<C>(q.y);
                    acc = sj_add_aff<C>(acc, q);
                }
            }
        }
    } else {
        // signed digits of u2: with e = u2 + 0x88..8, digit i = nibble i of e - 8 in [-8, 7] (i < 64), digit 64 = the carry
        Fr e;
        const uint32_t top = fr_add256(e, u2, eights);
        if (top) acc = SJac{Q.x, Q.y, secp_one()};
#pragma unroll 1
        for (int i = 255; i >= 0; i--) {  // one doubling and one addition in the loop body: the code stays within reach of the instruction cache
            acc = sj_dbl<C>(acc);
            if (i & 3) continue;
            const int32_t dg = (int32_t)((secp_limb_at(e, (uint32_t)i >> 5) >> (i & 31)) & 15u) - 8;
            if (dg != 0) {
                const uint32_t mag = (uint32_t)(dg < 0 ? -dg : dg);
                SAff q = tab[mag - 1u];
                if (dg < 0) q.y = sp_neg<C>(q.y);
                acc = sj_add_aff<C>(acc, q);
            }
        }
    }
    return secp_add_generator_multiple<C>(acc, u1, gtab);
}

// panic codes (host texts in batch.cpp ecdsa_panic_text)
enum EcdsaPanic : uint32_t { EP_SIG = 1, EP_PUBKEY = 2, EP_MSG_LEN = 3, EP_MSG_RANGE = 4, EP_IDENTITY = 5, EP_X_RANGE = 6 };

// The verification proper on parsed integers, in the order of the reference's checks (ops_ecdsa.hpp): 1 valid, 0 invalid, or a panic code in *panic.
// gtab: this curve's table of the generator.
// y_given (optional): the 32 bytes of public_key_y as an integer. The reference decompresses the key from x and the parity of y's last byte
// (from_affine_coordinates(.., compress = true)): the root of x^3 + a x + b with that parity. When the given y is a field element that
// squares to the right-hand side it IS that root (the other root p - y has the other parity), and the 253 squarings of the square root --
// 8 % / 12 % of a secp256r1 / secp256k1 verification -- are skipped; an off-curve or unreduced y takes the square root as before.
template <int C>
FR_HD inline __noinline__ uint32_t secp_verify(const Fr &r, const Fr &s, const Fr &x, uint32_t y_odd, uint32_t n_msg, const Fr &z,
                                              const uint32_t *__restrict__ gtab, uint32_t *panic, const Fr *y_given = nullptr) {
    *panic = 0;
    const Fr n = sn_modulus<C>();
    if (fr_is_zero(r) || fr_is_zero(s) || secp_geq(r, n) || secp_geq(s, n)) { *panic = EP_SIG; return 0; }
    if (secp_geq(x, sp_modulus<C>())) { *panic = EP_PUBKEY; return 0; }
    Fr rhs = sp_mul<C>(sp_sqr<C>(x), x);
    if (C == 1) rhs = sp_sub<C>(rhs, sp_add<C>(sp_add<C>(x, x), x));
    rhs = sp_add<C>(rhs, secp_limbs([](int i) { return Secp<C>::b(i); }));
    Fr y = fr_zero();
    bool have_y = false;
    if (y_given && ((y_given->v[0] ^ y_odd) & 1u) == 0u && !secp_geq(*y_given, sp_modulus<C>()) && fr_eq(sp_sqr<C>(*y_given), rhs)) {
        y = *y_given;
        have_y = true;
    }
    if (!have_y) {
        y = sp_sqrt_candidate<C>(rhs);
        if (!fr_eq(sp_sqr<C>(y), rhs)) { *panic = EP_PUBKEY; return 0; }
        if ((y.v[0] & 1u) != (y_odd & 1u)) y = sp_neg<C>(y);
    }
    if (n_msg != 32u) { *panic = EP_MSG_LEN; return 0; }
    if (secp_geq(z, n)) { *panic = EP_MSG_RANGE; return 0; }
    Fr d;
    if (fr_sub256(d, secp_limbs([](int i) { return Secp<C>::half_n(i); }), s)) return 0;  // s > n / 2: not low-S normalised
    const Fr si = sn_inv<C>(s);
    const Fr u1 = sn_mul<C>(z, si), u2 = sn_mul<C>(r, si);
    const SJac R = secp_mul2<C>(u1, SAff{x, y}, u2, gtab);
    if (fr_is_zero(R.Z)) { *panic = EP_IDENTITY; return 0; }
    const Fr zi = sp_inv<C>(R.Z);
    const Fr rx = sp_mul<C>(R.X, sp_sqr<C>(zi));
    if (secp_geq(rx, n)) { *panic = EP_X_RANGE; return 0; }
    return fr_eq(rx, r) ? 1u : 0u;
}

}  // namespace acvm
