// secp_device.hpp -- arithmetic of secp256k1 / secp256r1 for the ECDSA opcodes (ops_ecdsa.hpp), one lane per verification.
//   blackbox_solver/src/lib.rs:66-210 (verify_secp256k1/r1_ecdsa_signature; k256 0.11.6 / p256 0.11.1 = SEC 1 v2 section 4.1.4).
// Everything is __host__ __device__ with compile-time moduli, so tools/secp_device_host_test.hip runs the same code on the host against
// Python integers (tests/test_secp_device_on_host.py).
//   * base field: 9 x 29-bit limbs in 32-bit registers with unreduced sums between products (the "working form" section below): plain residues
//     folded through the shape of the prime for secp256k1 (2^261 = 2^37 + 31264, 2^256 = 2^32 + 977), Montgomery residues for secp256r1, whose
//     p = -1 (mod 2^29) makes the reduction multiplier the column's own low limb. (Round 3: 8 x 32-bit limbs with carry chains, 1.8 x the instructions.)
//   * scalar field (three products per verification): Montgomery, modulus a compile-time constant.
//   * inversions: safegcd (fr_device.hpp fr_safegcd_inv) with the modulus as a template parameter.
//   * square root for the decompression of the public key (both p = 3 mod 4): addition chains for (p + 1) / 4 (253 squarings + 13 / 7 products).
//   * u1 G + u2 Q: u2 Q on signed 4-bit windows over a per-lane table {Q .. 8Q} normalised to affine with ONE inversion (256 doublings +
//     65 mixed additions, every lane adds at the same steps); u1 G as 16 mixed additions from a precomputed table d * 2^(16 j) * G
//     (16 x 65 535 affine points per curve, 64 MiB, built once per device by secp_gtable_entry) onto the same accumulator.
#pragma once
#include "fr_device.hpp"
#if defined(SECP_CHECK)
#include <cstdio>
#include <cstdlib>
#endif

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

// ---- wide product of two 8 x 32-bit integers (the endomorphism split below), column by column: a column's 64-bit products are summed in a
// 96-bit accumulator (acc, top). On the device one term is v_mad_u64_u32 (the 64-bit accumulate of the multiplier, carry out in vcc) +
// v_addc_co_u32.
FR_HD __forceinline__ void secp_mac(uint64_t &acc, uint32_t &top, uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    // (the carry is named as the 64-bit vcc of a wave64 target: this library is built for gfx950 only, build.py)
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

// ---- base field, working form (round 4): 9 x 29-bit limbs in 32-bit registers (the container of fr_device.hpp, Fr29), like the BN254 code.
// Round 3 multiplied 8 x 32-bit limbs: 64 multiplies, but every one of them with a carry (v_addc), a 96-bit column accumulator that had to be
// shifted through registers (160 v_mov per product pair) and a carry-chained reduction: 290 (secp256k1) / 350 (secp256r1) instructions per
// product, and every instruction of this mix issues at the same rate (the back-to-back probes: 4 cycles each, v_mad_u64_u32 included). On
// 29-bit limbs a column of 9 products (< 2^58 each) sums in ONE 64-bit register with no carry at all: 81 multiplies + a mask and a shift per
// column, and the 5 spare bits (9 x 29 = 261) let sums and differences stay unreduced between products. ~165 instructions per product.
//   secp256k1: plain residues. 2^261 = 2^37 + 31264 (mod p) folds the high nine limbs of a product back with one multiply and one shift per
//              limb; what is then left above 2^256 folds with 2^256 = 2^32 + 977. A product is below 1.01 p whatever its operands were.
//   secp256r1: Montgomery residues x R, R = 2^261. p = -1 (mod 2^29), so the multiplier of a reduction step is the column's low limb itself,
//              and p = (2^96 - 1) + 2^192 + 2^224 (2^32 - 1) spreads it over four columns with two shifts and two multiplies: 36 instructions
//              for the whole reduction instead of the 81 multiplies of a generic modulus. A product of a < A p and b < B p is < (A B / 32 + 1) p.
// "Domain form" below = what the curve routines compute on: the residue (k1) or the Montgomery residue (r1); sp_enter / sp_leave convert.
// Bounds are written beside the formulas in units of p; "normalised" = limbs below 2^29 (the top one holds the rest).
using S29 = Fr29;
constexpr uint32_t S29_M = 0x1fffffffu;
// The bounds written beside the formulas are CHECKED when this header is compiled for the host with -DSECP_CHECK (tests/test_secp_device_on_host.py
// does): no limb-wise difference goes negative, no subtrahend exceeds its multiple of p, no column of a product leaves 64 bits, nothing above 128 p is folded.
#if defined(SECP_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
#define S29_ASSERT(c) do { if (!(c)) { fprintf(stderr, "secp_device.hpp:%d: bound violated: %s\n", __LINE__, #c); abort(); } } while (0)
#define S29_CHECKED 1
#else
#define S29_ASSERT(c) ((void)0)
#define S29_CHECKED 0
#endif

// The limbs of a result pass through an empty asm on the device: hipcc (ROCm 7.2, -O3, gfx950) otherwise carries what it knows about their widths into
// the NEXT product and miscompiles it -- a square of a square came out wrong on the device in 4 077 of 4 096 cases and right on the host. tools/narrow_probe.hip
// reproduces it standalone (profiles/r04_narrow_probe.txt): the limb that matters is limb 8, which the fold at 2^256 masks to 24 bits; laundering that one
// limb cures every case, laundering the other eight none. The library launders all nine (no instruction is emitted), and the compositions stay in the GPU
// suite (tests/test_gpu_secp_probe.py, probes 8..11).
FR_HD __forceinline__ void s29_opaque(S29 &r) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(S29_PROBE_NO_OPAQUE)  // (tools/narrow_probe.hip reproduces the miscompilation without it)
#pragma unroll
    for (int i = 0; i < 9; i++) asm volatile("" : "+v"(r.v[i]));
#else
    (void)r;
#endif
}
template <int C>
FR_HD __forceinline__ uint32_t sp29_p(int i) {
    constexpr uint32_t L[2][9] = {{0x1ffffc2fu, 0x1ffffff7u, 0x1fffffffu, 0x1fffffffu, 0x1fffffffu, 0x1fffffffu, 0x1fffffffu, 0x1fffffffu, 0x00ffffffu},
                                  {0x1fffffffu, 0x1fffffffu, 0x1fffffffu, 0x000001ffu, 0x00000000u, 0x00000000u, 0x00040000u, 0x1fe00000u, 0x00ffffffu}};
    return L[C][i];
}
// 2^klog2 p (klog2 = 0..4) with every limb below the top raised by 2^30 and the two borrowed from the next one: limb-wise a + K - b never goes
// negative below the top limb for limbs of b up to 2^30 - 2 (a normalised value or the sum of two); the top limb may wrap, which cancels in s29_norm
template <int C>
FR_HD __forceinline__ uint32_t sp29_kp_sub(int klog2, int i) {
    constexpr uint32_t L[2][5][9] = {
        {{0x5ffffc2fu, 0x5ffffff5u, 0x5ffffffdu, 0x5ffffffdu, 0x5ffffffdu, 0x5ffffffdu, 0x5ffffffdu, 0x5ffffffdu, 0x00fffffdu},
         {0x5ffff85eu, 0x5fffffedu, 0x5ffffffdu, 0x5ffffffdu, 0x5ffffffdu, 0x5ffffffdu, 0x5ffffffdu, 0x5ffffffdu, 0x01fffffdu},
         {0x5ffff0bcu, 0x5fffffddu, 0x5ffffffdu, 0x5ffffffdu, 0x5ffffffdu, 0x5ffffffdu, 0x5ffffffdu, 0x5ffffffdu, 0x03fffffdu},
         {0x5fffe178u, 0x5fffffbdu, 0x5ffffffdu, 0x5ffffffdu, 0x5ffffffdu, 0x5ffffffdu, 0x5ffffffdu, 0x5ffffffdu, 0x07fffffdu},
         {0x5fffc2f0u, 0x5fffff7du, 0x5ffffffdu, 0x5ffffffdu, 0x5ffffffdu, 0x5ffffffdu, 0x5ffffffdu, 0x5ffffffdu, 0x0ffffffdu}},
        {{0x5fffffffu, 0x5ffffffdu, 0x5ffffffdu, 0x400001fdu, 0x3ffffffeu, 0x3ffffffeu, 0x4003fffeu, 0x5fdffffeu, 0x00fffffdu},
         {0x5ffffffeu, 0x5ffffffdu, 0x5ffffffdu, 0x400003fdu, 0x3ffffffeu, 0x3ffffffeu, 0x4007fffeu, 0x5fbffffeu, 0x01fffffdu},
         {0x5ffffffcu, 0x5ffffffdu, 0x5ffffffdu, 0x400007fdu, 0x3ffffffeu, 0x3ffffffeu, 0x400ffffeu, 0x5f7ffffeu, 0x03fffffdu},
         {0x5ffffff8u, 0x5ffffffdu, 0x5ffffffdu, 0x40000ffdu, 0x3ffffffeu, 0x3ffffffeu, 0x401ffffeu, 0x5efffffeu, 0x07fffffdu},
         {0x5ffffff0u, 0x5ffffffdu, 0x5ffffffdu, 0x40001ffdu, 0x3ffffffeu, 0x3ffffffeu, 0x403ffffeu, 0x5dfffffeu, 0x0ffffffdu}}};
    return L[C][klog2][i];
}
FR_HD __forceinline__ S29 s29_zero() {
    S29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = 0u;
    return r;
}
FR_HD __forceinline__ S29 s29_norm(const S29 &a) { return fr29_norm(a); }   // carry propagation: limbs < 2^29 below the top one
FR_HD __forceinline__ S29 s29_addl(const S29 &a, const S29 &b) { return fr29_addl(a, b); }
FR_HD __forceinline__ S29 s29_dbll(const S29 &a) { return fr29_dbll(a); }
// a + 2^klog2 p - b, limb-wise: limbs of b <= 2^30 - 2, value(b) <= 2^klog2 p; limbs of a < 2^31. To be normalised before anything reads the top limb.
template <int C>
FR_HD __forceinline__ S29 s29_subl(const S29 &a, const S29 &b, int klog2) {
    S29 r;
#if S29_CHECKED
    {
        for (int i = 0; i < 8; i++) S29_ASSERT((uint64_t)a.v[i] + sp29_kp_sub<C>(klog2, i) >= b.v[i] && (uint64_t)a.v[i] + sp29_kp_sub<C>(klog2, i) - b.v[i] < (1ull << 32));
        // value(b) <= 2^klog2 p: compare the normalised limbs from the top
        uint64_t bn[9], kn[9], cb = 0, ck = 0;
        for (int i = 0; i < 9; i++) {
            cb += b.v[i];
            ck += (uint64_t)sp29_p<C>(i) << klog2;
            bn[i] = i < 8 ? (cb & S29_M) : cb;
            kn[i] = i < 8 ? (ck & S29_M) : ck;
            cb = i < 8 ? cb >> 29 : 0;
            ck = i < 8 ? ck >> 29 : 0;
        }
        int cmp = 0;
        for (int i = 8; i >= 0 && !cmp; i--) cmp = bn[i] < kn[i] ? -1 : bn[i] > kn[i] ? 1 : 0;
        S29_ASSERT(cmp <= 0);
    }
#endif
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + sp29_kp_sub<C>(klog2, i) - b.v[i];
    return r;
}
// normalised limbs, value < 128 p  ->  value < 2^256 + 2^232 (< 1.01 p), limbs < 2^29 (+ 1): what stands above bit 256 comes back through 2^256 mod p
template <int C>
FR_HD __forceinline__ S29 s29_weak(const S29 &a);
template <>
FR_HD __forceinline__ S29 s29_weak<0>(const S29 &a) {  // 2^256 = 2^32 + 977
    S29 r = a;
    const uint32_t e = r.v[8] >> 24;  // < 2^7
#if S29_CHECKED
    S29_ASSERT(e < 128u);
    for (int i = 0; i < 8; i++) S29_ASSERT(a.v[i] <= S29_M);
#endif
    r.v[8] &= 0x00ffffffu;
    r.v[0] += e * 977u;
    r.v[1] += e << 3;
#pragma unroll
    for (int i = 0; i < 3; i++) {  // (the carry that reaches limb 3 is 0 or 1: it stays there)
        r.v[i + 1] += r.v[i] >> 29;
        r.v[i] &= S29_M;
    }
    s29_opaque(r);
    return r;
}
template <>
FR_HD __forceinline__ S29 s29_weak<1>(const S29 &a) {  // 2^256 = 2^224 - 2^192 - 2^96 + 1; the two negative terms borrow from a raised zero (limbs 3..7)
    S29 r = a;
    const uint32_t e = r.v[8] >> 24, nz = e ? 0xffffffffu : 0u;  // e < 2^7
#if S29_CHECKED
    S29_ASSERT(e < 128u);
    for (int i = 0; i < 8; i++) S29_ASSERT(a.v[i] <= S29_M);
#endif
    r.v[8] &= 0x00ffffffu;
    r.v[0] += e;
    r.v[3] += (nz & 0x20000000u) - (e << 9);
    r.v[4] += nz & S29_M;
    r.v[5] += nz & S29_M;
    r.v[6] += (nz & S29_M) - (e << 18);
    r.v[7] += (e << 21) - (nz & 1u);
    r = s29_norm(r);
    s29_opaque(r);
    return r;
}
// normalised limbs, value < 2 p -> [0, p)
template <int C>
FR_HD __forceinline__ S29 s29_csub_p(const S29 &a) {
    S29 d;
    int32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int32_t t = (int32_t)a.v[i] - (int32_t)sp29_p<C>(i) + borrow;
        d.v[i] = i < 8 ? ((uint32_t)t & S29_M) : (uint32_t)t;
        borrow = i < 8 ? (t >> 29) : (t >> 31);
    }
    S29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = borrow ? a.v[i] : d.v[i];
    return r;
}
// limbs < 2^31 below the top, value < 128 p -> the canonical representative of the domain form
template <int C>
FR_HD __forceinline__ S29 s29_canon(const S29 &a) { return s29_csub_p<C>(s29_norm(s29_weak<C>(s29_norm(a)))); }
template <int C>
FR_HD __forceinline__ bool s29_is_zero(const S29 &a) {  // a as for s29_canon: is it 0 (mod p)
    const S29 c = s29_canon<C>(a);
    uint32_t z = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) z |= c.v[i];
    return z == 0u;
}
// the same for a coordinate as the point formulas keep them: exact limbs (secp256k1: limb 3 may read 2^29, which neither 0 nor p does), value < 2 p:
// it is 0 or p
template <int C>
FR_HD __forceinline__ bool s29_is_zero_coord(const S29 &a) {
#if S29_CHECKED
    for (int i = 0; i < 8; i++) S29_ASSERT(a.v[i] <= S29_M + (C == 0 && i == 3 ? 1u : 0u));
    S29_ASSERT(a.v[8] < (1u << 25));  // < 2 p
#endif
    uint32_t z = 0, e = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        z |= a.v[i];
        e |= a.v[i] ^ sp29_p<C>(i);
    }
    return z == 0u || e == 0u;
}
template <int C>
FR_HD __forceinline__ bool s29_eq(const S29 &a, const S29 &b) {  // a = b (mod p)
    const S29 x = s29_canon<C>(a), y = s29_canon<C>(b);
    uint32_t z = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) z |= x.v[i] ^ y.v[i];
    return z == 0u;
}

// ---- products. Operands: limbs < 2^30 (a normalised value, or the limb-wise sum of two), top limb < 2^31; a column of nine products stays
// below 2^64. secp256k1: any value; result < 1.01 p, limbs < 2^29 except limb 3 <= 2^29. secp256r1: result < (A B / 32 + 1) p, normalised.
#if S29_CHECKED
static inline void s29_check_columns(const S29 &a, const S29 &b) {  // every column of a b, with its carry, within 64 bits; the product's top within 32
    unsigned __int128 acc = 0;
    for (int k = 0; k < 17; k++) {
        for (int i = 0; i < 9; i++)
            if (k - i >= 0 && k - i < 9) acc += (unsigned __int128)a.v[i] * b.v[k - i];
        S29_ASSERT(acc < ((unsigned __int128)1 << 63));  // (room for the reduction terms of secp256r1)
        acc >>= 29;
    }
    S29_ASSERT(acc < ((unsigned __int128)1 << 32));
}
#endif
// a power of two the compiler must multiply by: a shift by more than 4 and the 64-bit addition behind it are two instructions, v_mad_u64_u32 is one
FR_HD __forceinline__ uint32_t s29_pow2(uint32_t log2) {
    uint32_t k = 1u << log2;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(k));
#endif
    return k;
}
// secp256k1. The HIGH nine columns of the product come first (h: 29 bits each, h[8] the rest; no carry-in from the low half, which joins at 2^261
// below); the low columns then take h (31264 + 2^8 2^29) as they are summed, so no limb of the wide product is formed twice.
//   `hi(k, acc)` adds column k of the product to acc (k = 0..16).
// ADD (round 6): a lazy sum `addend` (limbs below 2^32, any value below 64 p -- a difference K - b, a few of them added) joins the low columns as they are
// summed: result = a b + addend (mod p) in the same form as a plain product (below 1.01 p, exact limbs), with the carries of the addend propagated by
// the scan. It replaces s29_out(s29_subl(product, b)) -- two limb-wise passes, a carry pass and a fold, 54-72 instructions -- by 9 + the 9 of K - b.
template <bool ADD, class Column>
FR_HD __forceinline__ S29 s29_fold_k1(Column column, const S29 *addend = nullptr) {
    const uint32_t k256 = s29_pow2(8);
    uint32_t h[9];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 9; k < 17; k++) {
        acc = column(k, acc);
        h[k - 9] = (uint32_t)acc & S29_M;
        acc >>= 29;
    }
    h[8] = (uint32_t)acc;
    S29 r;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        c = column(i, c);
        c += (uint64_t)h[i] * 31264u;
        if (i > 0) c += (uint64_t)h[i - 1] * k256;
        if (ADD) c += addend->v[i];
        r.v[i] = (uint32_t)c & S29_M;
        c >>= 29;
    }
    c += (uint64_t)h[8] * k256;  // what stands at 2^261 now (< 2^42): once more through 31264 + 2^8 2^29, together with bits 256..260 through 977 + 2^3 2^29
    const uint32_t e = r.v[8] >> 24;
    r.v[8] &= 0x00ffffffu;
    uint64_t d = c * 31264u + (uint64_t)(e * 977u) + r.v[0];
    r.v[0] = (uint32_t)d & S29_M;
    d >>= 29;
    d += (c << 8) + (uint64_t)(e << 3) + r.v[1];
    r.v[1] = (uint32_t)d & S29_M;
    d >>= 29;
    d += r.v[2];
    r.v[2] = (uint32_t)d & S29_M;
    r.v[3] += (uint32_t)(d >> 29);  // 0 or 1
    s29_opaque(r);
    return r;
}
template <int C>
FR_HD __forceinline__ S29 s29_mul(const S29 &a, const S29 &b);
template <int C>
FR_HD __forceinline__ S29 s29_sqr(const S29 &a);
template <>
FR_HD __forceinline__ S29 s29_mul<0>(const S29 &a, const S29 &b) {
#if S29_CHECKED
    s29_check_columns(a, b);
#endif
    return s29_fold_k1<false>([&](int k, uint64_t acc) {
#pragma unroll
        for (int i = 0; i < 9; i++)
            if (k - i >= 0 && k - i < 9) acc += (uint64_t)a.v[i] * b.v[k - i];
        return acc;
    });
}
// a b + addend, a^2 + addend (see s29_fold_k1): secp256k1 here, secp256r1 below the plain products
template <int C>
FR_HD __forceinline__ S29 s29_mul_add(const S29 &a, const S29 &b, const S29 &addend);
template <int C>
FR_HD __forceinline__ S29 s29_sqr_add(const S29 &a, const S29 &addend);
#if S29_CHECKED
static inline void s29_check_addend(const S29 &addend) {  // limbs are 32-bit by type (the column accumulators have room for them); the value stays below 128 p
    S29_ASSERT(addend.v[8] < (1u << 31));                 // (value < (v8 + 16) 2^232: the lower limbs, below 2^32 each, add less than 2^236)
}
#endif
template <>
FR_HD __forceinline__ S29 s29_mul_add<0>(const S29 &a, const S29 &b, const S29 &addend) {
#if S29_CHECKED
    s29_check_columns(a, b);
    s29_check_addend(addend);
#endif
    return s29_fold_k1<true>([&](int k, uint64_t acc) {
#pragma unroll
        for (int i = 0; i < 9; i++)
            if (k - i >= 0 && k - i < 9) acc += (uint64_t)a.v[i] * b.v[k - i];
        return acc;
    }, &addend);
}
template <>
FR_HD __forceinline__ S29 s29_sqr<0>(const S29 &a) {  // the 36 cross products once, against the doubled limbs
#if S29_CHECKED
    s29_check_columns(a, a);
    for (int i = 0; i < 9; i++) S29_ASSERT(a.v[i] < (1u << 31));
#endif
    uint32_t d[9];
#pragma unroll
    for (int i = 0; i < 9; i++) d[i] = a.v[i] << 1;
    return s29_fold_k1<false>([&](int k, uint64_t acc) {
#pragma unroll
        for (int i = 0; i < 9; i++)
            if (k - i > i && k - i < 9) acc += (uint64_t)d[i] * a.v[k - i];
        if ((k & 1) == 0) acc += (uint64_t)a.v[k / 2] * a.v[k / 2];
        return acc;
    });
}
template <>
FR_HD __forceinline__ S29 s29_sqr_add<0>(const S29 &a, const S29 &addend) {
#if S29_CHECKED
    s29_check_columns(a, a);
    s29_check_addend(addend);
    for (int i = 0; i < 9; i++) S29_ASSERT(a.v[i] < (1u << 31));
#endif
    uint32_t d[9];
#pragma unroll
    for (int i = 0; i < 9; i++) d[i] = a.v[i] << 1;
    return s29_fold_k1<true>([&](int k, uint64_t acc) {
#pragma unroll
        for (int i = 0; i < 9; i++)
            if (k - i > i && k - i < 9) acc += (uint64_t)d[i] * a.v[k - i];
        if ((k & 1) == 0) acc += (uint64_t)a.v[k / 2] * a.v[k / 2];
        return acc;
    }, &addend);
}
// secp256r1: column k of a b + sum m_i p 2^(29 i). m_k = the low limb of column k; it leaves that column (the shift drops it) and enters
// columns k + 3 (2^9), k + 6 (2^18), k + 7 (0x1fe00000), k + 8 (0xffffff): p = (2^96 - 1) + 2^192 + 2^224 (2^32 - 1)
#define SECP_R1_REDUCE_STEP(k)                                                \
    if ((k) >= 3 && (k) - 3 < 9) acc += (uint64_t)m[(k) - 3] * k9;            \
    if ((k) >= 6 && (k) - 6 < 9) acc += (uint64_t)m[(k) - 6] * k18;           \
    if ((k) >= 7 && (k) - 7 < 9) acc += (uint64_t)m[(k) - 7] * 0x1fe00000u;   \
    if ((k) >= 8 && (k) - 8 < 9) acc += (uint64_t)m[(k) - 8] * 0x00ffffffu;   \
    if (ADD && (k) >= 9) acc += addend->v[(k) - 9];                           \
    if ((k) < 9) m[k] = (uint32_t)acc & S29_M;                                \
    else if ((k) < 17) r.v[(k) - 9] = (uint32_t)acc & S29_M;                  \
    acc >>= 29;
// ADD (round 6): a lazy sum rides in the upper columns (like fr29_dot_add of the BN254 code): result = (a b + m p) / 2^261 + addend exactly, normalised
// limbs, value < (A B / 32 + 1) p + addend -- the caller folds it with s29_weak<1> where it must be a coordinate (no separate carry pass)
template <bool ADD>
FR_HD __forceinline__ S29 s29_mul_r1(const S29 &a, const S29 &b, const S29 *addend) {
#if S29_CHECKED
    s29_check_columns(a, b);
#endif
    const uint32_t k9 = s29_pow2(9), k18 = s29_pow2(18);
    uint32_t m[9];
    S29 r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 17; k++) {
#pragma unroll
        for (int i = 0; i < 9; i++)
            if (k - i >= 0 && k - i < 9) acc += (uint64_t)a.v[i] * b.v[k - i];
        SECP_R1_REDUCE_STEP(k)
    }
    r.v[8] = (uint32_t)acc;
    if (ADD) r.v[8] += addend->v[8];
    s29_opaque(r);
    return r;
}
template <bool ADD>
FR_HD __forceinline__ S29 s29_sqr_r1(const S29 &a, const S29 *addend) {
#if S29_CHECKED
    s29_check_columns(a, a);
    for (int i = 0; i < 9; i++) S29_ASSERT(a.v[i] < (1u << 31));
#endif
    const uint32_t k9 = s29_pow2(9), k18 = s29_pow2(18);
    uint32_t m[9], d[9];
#pragma unroll
    for (int i = 0; i < 9; i++) d[i] = a.v[i] << 1;
    S29 r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 17; k++) {
#pragma unroll
        for (int i = 0; i < 9; i++)
            if (k - i > i && k - i < 9) acc += (uint64_t)d[i] * a.v[k - i];
        if ((k & 1) == 0) acc += (uint64_t)a.v[k / 2] * a.v[k / 2];
        SECP_R1_REDUCE_STEP(k)
    }
    r.v[8] = (uint32_t)acc;
    if (ADD) r.v[8] += addend->v[8];
    s29_opaque(r);
    return r;
}
template <>
FR_HD __forceinline__ S29 s29_mul<1>(const S29 &a, const S29 &b) { return s29_mul_r1<false>(a, b, nullptr); }
template <>
FR_HD __forceinline__ S29 s29_sqr<1>(const S29 &a) { return s29_sqr_r1<false>(a, nullptr); }
template <>
FR_HD __forceinline__ S29 s29_mul_add<1>(const S29 &a, const S29 &b, const S29 &addend) { return s29_mul_r1<true>(a, b, &addend); }
template <>
FR_HD __forceinline__ S29 s29_sqr_add<1>(const S29 &a, const S29 &addend) { return s29_sqr_r1<true>(a, &addend); }
#undef SECP_R1_REDUCE_STEP
template <int C>
FR_HD __forceinline__ S29 s29_sqr_n(S29 a, int n) {
#pragma unroll 1
    for (int i = 0; i < n; i++) a = s29_sqr<C>(a);
    return a;
}

// ---- domain form <-> canonical integers (8 x 32)
FR_HD __forceinline__ S29 sp_load(const Fr &a) {
    S29 r = fr29_from(a);
    s29_opaque(r);
    return r;
}
template <int C>
FR_HD __forceinline__ S29 sp_one() {  // 1 in the domain form
    if (C == 0) {
        S29 o = s29_zero();
        o.v[0] = 1u;
        return o;
    }
    return fr29_from(Fr{{0x00000020u, 0x00000000u, 0x00000000u, 0xffffffe0u, 0xffffffffu, 0xffffffffu, 0xffffffdfu, 0x0000001fu}});  // R mod p
}
template <int C>
FR_HD __forceinline__ S29 sp_enter(const Fr &x) {  // x < 2^256
    if (C == 0) return sp_load(x);
    return s29_mul<1>(fr29_from(x), fr29_from(Fr{{0x00000c00u, 0x00000000u, 0xfffffc00u, 0xffffefffu, 0xfffffbffu, 0xffffffffu, 0xfffff7ffu, 0x000013ffu}}));  // R^2 mod p
}
template <int C>
FR_HD __forceinline__ Fr sp_leave(const S29 &a) {  // a as for s29_canon -> [0, p)
    if (C == 0) return fr29_pack(s29_canon<0>(a));
    S29 one = s29_zero();
    one.v[0] = 1u;
    return fr29_pack(s29_canon<1>(s29_mul<1>(s29_norm(s29_weak<1>(s29_norm(a))), one)));
}
// the canonical limbs of a domain value, as the tables hold them (no change of form)
template <int C>
FR_HD __forceinline__ Fr sp_store(const S29 &a) { return fr29_pack(s29_canon<C>(a)); }
template <int C>
FR_HD inline __noinline__ S29 sp29_inv(const S29 &a) {  // 1 / a in the domain form; 0 for 0
    const Fr i = fr_safegcd_inv<SecpMod30<2 * C>>(sp_store<C>(a));
    if (C == 0) return fr29_from(i);
    // (a R)^-1 R^3 / R = a^-1 R
    return s29_mul<1>(fr29_from(i), fr29_from(Fr{{0x00050000u, 0xfffe8000u, 0xfffbffffu, 0xfff6ffffu, 0xfffe7fffu, 0x0002ffffu, 0x00008000u, 0x000c0000u}}));
}
template <int C>
FR_HD inline __noinline__ Fr sn_inv(const Fr &a) { return fr_safegcd_inv<SecpMod30<2 * C + 1>>(a); }

// a^((p + 1) / 4): the square root of a when a is a square (both primes are 3 mod 4). Operand < 2 p, normalised; result < 2 p.
template <int C>
FR_HD inline __noinline__ S29 sp29_sqrt_candidate(const S29 &a);
// (p + 1) / 4 = 1{223} 0 1{22} 0000 11 00 in binary (the chain of libsecp256k1's field square root)
template <>
FR_HD inline __noinline__ S29 sp29_sqrt_candidate<0>(const S29 &a) {
    const S29 x2 = s29_mul<0>(s29_sqr<0>(a), a), x3 = s29_mul<0>(s29_sqr<0>(x2), a);
    const S29 x6 = s29_mul<0>(s29_sqr_n<0>(x3, 3), x3), x9 = s29_mul<0>(s29_sqr_n<0>(x6, 3), x3), x11 = s29_mul<0>(s29_sqr_n<0>(x9, 2), x2);
    const S29 x22 = s29_mul<0>(s29_sqr_n<0>(x11, 11), x11), x44 = s29_mul<0>(s29_sqr_n<0>(x22, 22), x22), x88 = s29_mul<0>(s29_sqr_n<0>(x44, 44), x44);
    const S29 x176 = s29_mul<0>(s29_sqr_n<0>(x88, 88), x88), x220 = s29_mul<0>(s29_sqr_n<0>(x176, 44), x44), x223 = s29_mul<0>(s29_sqr_n<0>(x220, 3), x3);
    S29 t = s29_mul<0>(s29_sqr_n<0>(x223, 23), x22);
    t = s29_mul<0>(s29_sqr_n<0>(t, 6), x2);
    return s29_sqr_n<0>(t, 2);
}
// (p + 1) / 4 = (2^32 - 1) 2^222 + 2^190 + 2^94
template <>
FR_HD inline __noinline__ S29 sp29_sqrt_candidate<1>(const S29 &a) {
    const S29 x2 = s29_mul<1>(s29_sqr<1>(a), a), x4 = s29_mul<1>(s29_sqr_n<1>(x2, 2), x2), x8 = s29_mul<1>(s29_sqr_n<1>(x4, 4), x4);
    const S29 x16 = s29_mul<1>(s29_sqr_n<1>(x8, 8), x8), x32 = s29_mul<1>(s29_sqr_n<1>(x16, 16), x16);
    S29 t = s29_mul<1>(s29_sqr_n<1>(x32, 32), a);
    t = s29_mul<1>(s29_sqr_n<1>(t, 96), a);
    return s29_sqr_n<1>(t, 94);
}

// ---- the same field operations on canonical integers (the host-run test drives these; the curve routines stay in the working form)
template <int C> FR_HD __forceinline__ Fr sp_mul(const Fr &a, const Fr &b) { return sp_leave<C>(s29_mul<C>(sp_enter<C>(a), sp_enter<C>(b))); }
template <int C> FR_HD __forceinline__ Fr sp_sqr(const Fr &a) { return sp_leave<C>(s29_sqr<C>(sp_enter<C>(a))); }
template <int C> FR_HD __forceinline__ Fr sp_add(const Fr &a, const Fr &b) { return sp_leave<C>(s29_addl(sp_enter<C>(a), sp_enter<C>(b))); }
template <int C> FR_HD __forceinline__ Fr sp_sub(const Fr &a, const Fr &b) { return sp_leave<C>(s29_subl<C>(sp_enter<C>(a), sp_enter<C>(b), 1)); }
template <int C> FR_HD __forceinline__ Fr sp_inv(const Fr &a) { return sp_leave<C>(sp29_inv<C>(sp_enter<C>(a))); }
template <int C> FR_HD __forceinline__ Fr sp_sqrt_candidate(const Fr &a) { return sp_leave<C>(sp29_sqrt_candidate<C>(sp_enter<C>(a))); }

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
// 200 000 scalars by the script quoted in NOTEBOOK.md section 9; tests/test_secp_device_on_host.py checks k1 + k2 lambda = k.)
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

// ---- points: Jacobian (X / Z^2, Y / Z^3) in the domain form. Coordinates: exact limbs below 2^29 (secp256k1: limb 3 may read 2^29), value
// < 1.1 p -- a product of small operands as it is, anything else through s29_out; Z = 0 (mod p) <=> identity; a = 0 (secp256k1) or a = -3 (secp256r1). Affine points (tables, inputs): canonical limbs.
struct SJac { S29 X, Y, Z; };
struct SAff { Fr x, y; };
template <int C>
FR_HD __forceinline__ SJac sj_identity() { return SJac{sp_one<C>(), sp_one<C>(), s29_zero()}; }
template <int C>
FR_HD __forceinline__ S29 s29_out(const S29 &lazy) { return s29_weak<C>(s29_norm(lazy)); }  // value < 128 p, limbs < 2^32 -> an output coordinate
// GUARDED = false (round 6, the ladders of secp_mul2): no test for the identity or a point of order two in front of the formulas. Both groups have prime
// order -- no finite point has Y = 0 -- and the formulas keep the identity the identity: Z3 = 2 Y Z is an exact zero whenever Z is (a product with a
// zero operand), whatever X and Y hold, and sj_add_aff29 looks at Z only. Two nine-limb tests and two branches fewer per doubling.
template <int C, bool GUARDED = true>
FR_HD __forceinline__ SJac sj_dbl(const SJac &p) {
    if (GUARDED && (s29_is_zero_coord<C>(p.Z) || s29_is_zero_coord<C>(p.Y))) return sj_identity<C>();
    SJac r;
    if (C == 0) {  // dbl-2009-l: 2 M + 5 S
        const S29 A = s29_sqr<C>(p.X), B = s29_sqr<C>(p.Y), Cc = s29_sqr<C>(B);                          // < 1.01
        const S29 s = s29_sqr<C>(s29_addl(p.X, B));                                                       // (X + B)^2
        const S29 t = s29_norm(s29_subl<C>(s29_subl<C>(s, A, 1), Cc, 1));                                 // < 5.01
        const S29 D = s29_dbll(t);                                                                        // < 10.02, limbs < 2^30
        const S29 E = s29_norm(s29_addl(s29_dbll(A), A));                                                 // 3 A < 3.03
        // (round 6: the differences ride in the products' column scans, s29_fold_k1 ADD: X3 = E^2 + (32 p - 2 D), Y3 = E (D - X3) + (16 p - 8 C), Z3 = (2 Y) Z --
        // each a product's output, i.e. a coordinate, with no carry pass and no fold of its own)
        r.X = s29_sqr_add<C>(E, s29_subl<C>(s29_subl<C>(s29_zero(), D, 4), D, 4));                        // E^2 - 2 D
        const S29 C8 = s29_dbll(s29_norm(s29_dbll(s29_dbll(Cc))));                                        // < 8.08, limbs <= 2^30 - 2
        r.Y = s29_mul_add<C>(E, s29_norm(s29_subl<C>(D, r.X, 1)), s29_subl<C>(s29_zero(), C8, 4));        // E (D - X3) - 8 C
        r.Z = s29_mul<C>(s29_dbll(p.Y), p.Z);
    } else {  // dbl-2001-b (a = -3): 3 M + 5 S
        const S29 delta = s29_sqr<C>(p.Z), gamma = s29_sqr<C>(p.Y);                                       // < 1.04
        const S29 beta = s29_mul<C>(p.X, gamma);                                                          // < 1.04
        const S29 u = s29_norm(s29_subl<C>(p.X, delta, 1));                                               // X - delta < 3.01
        const S29 al0 = s29_mul<C>(u, s29_addl(p.X, delta));                                              // (X - delta)(X + delta): 3.01 * 2.05 / 32 + 1 < 1.2
        const S29 alpha = s29_norm(s29_addl(s29_dbll(al0), al0));                                         // < 3.6
        const S29 b8 = s29_dbll(s29_norm(s29_dbll(s29_dbll(beta))));                                      // < 8.32, limbs <= 2^30 - 2
        r.X = s29_weak<C>(s29_sqr_add<C>(alpha, s29_subl<C>(s29_zero(), b8, 4)));                         // alpha^2 - 8 beta: the difference rides in the square's upper columns (round 6), < 17.5 before the fold
        r.Z = s29_mul<C>(s29_dbll(p.Y), p.Z);                                                             // (2 Y) Z as a product: 2.2 * 1.1 / 32 + 1 < 1.08 (the form (Y + Z)^2 - gamma - delta
                                                                                                          // pays two differences and a reduction for its squaring)
        const S29 w = s29_norm(s29_subl<C>(s29_dbll(s29_dbll(beta)), r.X, 1));                            // 4 beta - X3 < 6.16
        const S29 g8 = s29_dbll(s29_norm(s29_dbll(s29_dbll(s29_sqr<C>(gamma)))));                         // 8 gamma^2 < 8.3
        r.Y = s29_weak<C>(s29_mul_add<C>(alpha, w, s29_subl<C>(s29_zero(), g8, 4)));                      // alpha w (3.6 * 6.16 / 32 + 1 < 1.7) - 8 gamma^2, < 17.7 before the fold
    }
    return r;
}
// complete mixed addition of a finite affine point (x2, y2: domain form, exact limbs, <= p): 8 M + 3 S
template <int C>
FR_HD __forceinline__ SJac sj_add_aff29(const SJac &p, const S29 &x2, const S29 &y2) {
    if (s29_is_zero_coord<C>(p.Z)) return SJac{x2, y2, sp_one<C>()};
    // (round 6: every difference rides in a product's column scan -- s29_mul_add / s29_sqr_add -- instead of a limb-wise pass, a carry pass and a fold
    // behind the product. secp256k1: the product's output IS a coordinate; secp256r1: one s29_weak brings it back below 1.01 p.)
    const S29 z1z1 = s29_sqr<C>(p.Z);
    S29 h, rr;
    if (C == 0) {
        h = s29_mul_add<C>(x2, z1z1, s29_subl<C>(s29_zero(), p.X, 1));                                    // x2 Z^2 - X < 1.01
        rr = s29_mul_add<C>(s29_mul<C>(y2, p.Z), z1z1, s29_subl<C>(s29_zero(), p.Y, 1));                  // y2 Z^3 - Y < 1.01
    } else {
        h = s29_weak<C>(s29_mul_add<C>(x2, z1z1, s29_subl<C>(s29_zero(), p.X, 1)));                       // < 3.1 before the fold
        rr = s29_norm(s29_subl<C>(s29_mul<C>(s29_mul<C>(y2, p.Z), z1z1), p.Y, 1));                        // < 3.07 (a carry pass is cheaper than this curve's fold)
    }
    if (s29_is_zero_coord<C>(h)) return s29_is_zero<C>(rr) ? sj_dbl<C>(p) : sj_identity<C>();
    const S29 hh = s29_sqr<C>(h), hhh = s29_mul<C>(hh, h), v = s29_mul<C>(p.X, hh);                       // < 1.04
    SJac r;
    const S29 xs = s29_sqr_add<C>(rr, s29_subl<C>(s29_subl<C>(s29_zero(), hhh, 1), s29_dbll(v), 2));      // rr^2 - H^3 - 2 V (+ 6 p)
    r.X = C == 0 ? xs : s29_weak<C>(xs);
    const S29 ys = s29_mul_add<C>(rr, s29_norm(s29_subl<C>(v, r.X, 1)), s29_subl<C>(s29_zero(), s29_mul<C>(p.Y, hhh), 1));  // rr (V - X3) - Y H^3 (+ 2 p)
    r.Y = C == 0 ? ys : s29_weak<C>(ys);
    r.Z = s29_mul<C>(p.Z, h);
    return r;
}
template <int C>
FR_HD __forceinline__ SJac sj_add_aff(const SJac &p, const SAff &q) { return sj_add_aff29<C>(p, sp_load(q.x), sp_load(q.y)); }
template <int C>
FR_HD __forceinline__ SAff sj_to_affine(const SJac &p) {  // p finite; the canonical limbs of the domain form
    const S29 zi = sp29_inv<C>(p.Z), zi2 = s29_sqr<C>(zi);
    return SAff{sp_store<C>(s29_mul<C>(p.X, zi2)), sp_store<C>(s29_mul<C>(p.Y, s29_mul<C>(zi2, zi)))};
}
template <int C>
FR_HD __forceinline__ SAff secp_generator() {  // domain form
    if (C == 0) return SAff{secp_limbs([](int i) { return Secp<C>::gx(i); }), secp_limbs([](int i) { return Secp<C>::gy(i); })};
    return SAff{Fr{{0x15228783u, 0x3ce61a83u, 0xfdb6c02fu, 0xb752bf88u, 0xec44a20eu, 0x3f6e656eu, 0xa6eab8ccu, 0x120beed7u}},
                Fr{{0xd2aac150u, 0xbe4a6af9u, 0x433c8b9bu, 0x69571c87u, 0xa43e64b1u, 0x5d10d11bu, 0xb10bb0aau, 0xae3fe314u}}};
}

// ---- the table of the generator: entry (j, d) = d * 2^(W j) * G, j < 256 / W, 1 <= d < 2^W, 16 words (x, y; canonical limbs of the domain form) at
// (((j << W) + d) * 16). W = SECP_GWIN_BITS: 16 on the device (16 additions for u1 G, 64 MiB per curve); the host-run test builds its table with -DSECP_GWIN_BITS=8.
#ifndef SECP_GWIN_BITS
#define SECP_GWIN_BITS 16
#endif
constexpr uint32_t SECP_GWIN = SECP_GWIN_BITS, SECP_GWINDOWS = 256u / SECP_GWIN;
static_assert(SECP_GWIN == 8u || SECP_GWIN == 16u, "window of the generator table");
constexpr uint32_t SECP_GTABLE_WORDS = (SECP_GWINDOWS << SECP_GWIN) * 16u;  // per curve
template <int C>
FR_HD inline SAff secp_gtable_entry(uint32_t j, uint32_t d) {
    const SAff G = secp_generator<C>();
    SJac base{sp_load(G.x), sp_load(G.y), sp_one<C>()};
    for (uint32_t i = 0; i < SECP_GWIN * j; i++) base = sj_dbl<C>(base);
    const SAff B = sj_to_affine<C>(base);
    SJac acc = sj_identity<C>();
    for (int i = (int)SECP_GWIN - 1; i >= 0; i--) {
        acc = sj_dbl<C>(acc);
        if ((d >> i) & 1u) acc = sj_add_aff<C>(acc, B);
    }
    return sj_to_affine<C>(acc);
}

// ---- u1 G + u2 Q
// a row of the per-lane window table: an affine point in the working form (a lane indexes the table by its own digit: it lives in the lane's scratch)
struct SAff29 { S29 x, y; };
// tab <- {Q, 2Q, .., 8Q} affine: seven Jacobian multiples, their Z inverted together (Montgomery's trick: 3 products per point + one inversion)
template <int C>
FR_HD __forceinline__ void secp_window_table(SAff29 tab[8], const S29 &qx, const S29 &qy) {
    S29 zs[8], pre[8];
    tab[0] = SAff29{qx, qy};
    SJac cur = sj_dbl<C>(SJac{qx, qy, sp_one<C>()});
    tab[1] = SAff29{cur.X, cur.Y};
    zs[1] = pre[1] = cur.Z;
#pragma unroll 1
    for (int k = 2; k < 8; k++) {
        cur = sj_add_aff29<C>(cur, qx, qy);
        tab[k] = SAff29{cur.X, cur.Y};
        zs[k] = cur.Z;
        pre[k] = s29_mul<C>(pre[k - 1], cur.Z);
    }
    // k Q is finite for 1 <= k <= 8 (prime order > 8), so every Z is invertible
    S29 inv = sp29_inv<C>(pre[7]);
#pragma unroll 1
    for (int k = 7; k >= 1; k--) {
        const S29 zi = k > 1 ? s29_mul<C>(inv, pre[k - 1]) : inv;
        if (k > 1) inv = s29_mul<C>(inv, zs[k]);
        const S29 zi2 = s29_sqr<C>(zi);
        tab[k] = SAff29{s29_mul<C>(tab[k].x, zi2), s29_mul<C>(tab[k].y, s29_mul<C>(zi2, zi))};
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
// -y of a table row (y a product of small operands: exact limbs, < 1.1 p): 2 p - y through s29_out, so that it is a coordinate again
template <int C>
FR_HD __forceinline__ S29 s29_neg_row(const S29 &y) { return s29_out<C>(s29_subl<C>(s29_zero(), y, 1)); }
// Q = (qx, qy): domain form, canonical
template <int C>
FR_HD inline __noinline__ SJac secp_mul2(const Fr &u1, const S29 &qx, const S29 &qy, const Fr &u2, const uint32_t *__restrict__ gtab) {
    SAff29 tab[8];
    secp_window_table<C>(tab, qx, qy);
#ifdef SECP_EXP_NO_TABLE_READS  // measurement only (NOTEBOOK.md section 9): every addition takes Q itself from registers, the table is built and kept alive
#define SECP_ROW(k) SAff29{qx, qy}
#define SECP_BX(k) qx
#pragma unroll
    for (int k = 0; k < 8; k++) asm volatile("" ::"v"(tab[k].x.v[0]), "v"(tab[k].y.v[0]));
#else
#define SECP_ROW(k) tab[k]
#define SECP_BX(k) bx[k]
#endif
    Fr eights;
#pragma unroll
    for (int i = 0; i < 8; i++) eights.v[i] = 0x88888888u;
    SJac acc = sj_identity<C>();
    if constexpr (C == 0) {
        // u2 = k1 + k2 lambda: two 128-bit ladders on one accumulator, 128 doublings; lambda (k Q) = (beta x, y) of the same table
        const S29 beta = fr29_from(Fr{{0x719501eeu, 0xc1396c28u, 0x12f58995u, 0x9cf04975u, 0xac3434e9u, 0x6e64479eu, 0x657c0710u, 0x7ae96a2bu}});
        S29 bx[8];
#pragma unroll 1
        for (int k = 0; k < 8; k++) bx[k] = s29_mul<C>(tab[k].x, beta);
        const SecpSplit sp = secp256k1_split_lambda(u2);
        // signed digits: e = k + 0x88..8 over 32 nibbles, digit i = nibble i of e - 8 (i < 32), digit 32 = bit 128 of e
        Fr e1, e2, eights128 = eights;
#pragma unroll
        for (int i = 4; i < 8; i++) eights128.v[i] = 0u;
        fr_add256(e1, sp.k1, eights128);
        fr_add256(e2, sp.k2, eights128);
#pragma unroll 1
        for (int i = 128; i >= 0; i--) {
            if (i != 128) acc = sj_dbl<C, false>(acc);
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
                    SAff29 q = SECP_ROW(mag - 1u);
                    if (h) q.x = SECP_BX(mag - 1u);
                    if (dg < 0) q.y = s29_neg_row<C>(q.y);
                    acc = sj_add_aff29<C>(acc, q.x, q.y);
                }
            }
        }
    } else {
        // signed digits of u2: with e = u2 + 0x88..8, digit i = nibble i of e - 8 in [-8, 7] (i < 64), digit 64 = the carry
        Fr e;
        const uint32_t top = fr_add256(e, u2, eights);
        if (top) acc = SJac{qx, qy, sp_one<C>()};
#pragma unroll 1
        for (int i = 255; i >= 0; i--) {  // one doubling and one addition in the loop body: the code stays within reach of the instruction cache
            acc = sj_dbl<C, false>(acc);
            if (i & 3) continue;
            const int32_t dg = (int32_t)((secp_limb_at(e, (uint32_t)i >> 5) >> (i & 31)) & 15u) - 8;
            if (dg != 0) {
                const uint32_t mag = (uint32_t)(dg < 0 ? -dg : dg);
                SAff29 q = SECP_ROW(mag - 1u);
                if (dg < 0) q.y = s29_neg_row<C>(q.y);
                acc = sj_add_aff29<C>(acc, q.x, q.y);
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
    // x^3 + a x + b in the domain form
    const S29 xd = sp_enter<C>(x);
    S29 rhs = s29_mul<C>(s29_sqr<C>(xd), xd);                                                       // < 1.04
    if (C == 1) {
        const S29 x3 = s29_norm(s29_addl(s29_dbll(xd), xd));                                        // 3 x < 3.2
        const S29 bR = fr29_from(Fr{{0x3897bbfbu, 0x139bec45u, 0x1086121bu, 0x9e00b994u, 0xe425dad5u, 0xb444157eu, 0x90e90681u, 0x8600c3bbu}});  // b R mod p
        rhs = s29_out<C>(s29_addl(s29_subl<C>(rhs, x3, 2), bR));
    } else {
        S29 seven = s29_zero();
        seven.v[0] = 7u;
        rhs = s29_out<C>(s29_addl(rhs, seven));
    }
    S29 yd = s29_zero();
    bool have_y = false;
    if (y_given && ((y_given->v[0] ^ y_odd) & 1u) == 0u && !secp_geq(*y_given, sp_modulus<C>())) {
        yd = sp_enter<C>(*y_given);
        have_y = s29_eq<C>(s29_sqr<C>(yd), rhs);
    }
    if (!have_y) {
        yd = sp29_sqrt_candidate<C>(rhs);
        if (!s29_eq<C>(s29_sqr<C>(yd), rhs)) { *panic = EP_PUBKEY; return 0; }
        if ((sp_leave<C>(yd).v[0] & 1u) != (y_odd & 1u)) yd = s29_subl<C>(s29_zero(), yd, 1);     // (a product: < 2 p)
    }
    yd = sp_load(sp_store<C>(yd));  // canonical: a table row
    if (n_msg != 32u) { *panic = EP_MSG_LEN; return 0; }
    if (secp_geq(z, n)) { *panic = EP_MSG_RANGE; return 0; }
    Fr d;
    if (fr_sub256(d, secp_limbs([](int i) { return Secp<C>::half_n(i); }), s)) return 0;  // s > n / 2: not low-S normalised
    const Fr si = sn_inv<C>(s);
    const Fr u1 = sn_mul<C>(z, si), u2 = sn_mul<C>(r, si);
    const SJac R = secp_mul2<C>(u1, sp_load(sp_store<C>(xd)), yd, u2, gtab);
    if (s29_is_zero_coord<C>(R.Z)) { *panic = EP_IDENTITY; return 0; }
    const S29 zi = sp29_inv<C>(R.Z);
    const Fr rx = sp_leave<C>(s29_mul<C>(R.X, s29_sqr<C>(zi)));
    if (secp_geq(rx, n)) { *panic = EP_X_RANGE; return 0; }
    return fr_eq(rx, r) ? 1u : 0u;
}

}  // namespace acvm
