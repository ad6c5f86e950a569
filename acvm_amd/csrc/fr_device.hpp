// fr_device.hpp -- BN254-Fr arithmetic for gfx950 (CDNA4), the inner loop of every kernel.
// Replaces acir_field::FieldElement add/sub/neg/mul/inverse (acir_field/src/generic_ark.rs:242-245,
// 360-406; ark-ff Fp256<MontBackend<_,4>>).
// Storage form `Fr`: the Montgomery representative x * R mod p with R = 2^261, as 8 x 32-bit limbs. Everything a
// non-Arithmetic opcode, the inversion kernel or a caller reads is fully reduced to [0, p), so equality / is_zero are limb
// compares, matching the reference's canonical-bytes equality (generic_ark.rs:88-92,164-169); a row that only Arithmetic
// gates read may hold ANY representative of its residue below 2^256 (gate_eval.hpp "relaxed rows": the product tolerates
// it, and its readers outside the gate kernels multiply by the row's 1 / scale first, which reduces). Working form `Fr29`: 9 limbs of 29 bits (R = 2^(9*29)): CDNA4 has no 64x64
// multiplier, the native wide multiply-add is v_mad_u64_u32 (measured 4.6 cycles per wave64, tools/chainbench), and a
// carry-out costs another v_addc_co_u32 (3.6 cycles) per product with 32-bit limbs. With 29-bit limbs a whole column of
// a * b + m * p (<= 18 products of < 2^58) fits a 64-bit accumulator, so the product needs 162 multiply-adds and no
// carry instruction at all: 905 cycles per wave-product against 1 164 for the 8 x 32 product-scanning form with carries
// (tools/mul29bench). The seven spare bits of R also make the product tolerant of unreduced inputs (< 8p in, < 1.4p out).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FR_HD __host__ __device__

namespace acvm {

struct Fr {
    uint32_t v[8];
};

#define FR_P0 0xf0000001u
#define FR_P1 0x43e1f593u
#define FR_P2 0x79b97091u
#define FR_P3 0x2833e848u
#define FR_P4 0x8181585du
#define FR_P5 0xb85045b6u
#define FR_P6 0xe131a029u
#define FR_P7 0x30644e72u
#define FR_N0INV 0xefffffffu /* -p^-1 mod 2^32 */

FR_HD __forceinline__ uint32_t fr_p(int i) {
    constexpr uint32_t P[8] = {FR_P0, FR_P1, FR_P2, FR_P3, FR_P4, FR_P5, FR_P6, FR_P7};
    return P[i];
}

FR_HD __forceinline__ Fr fr_zero() {
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = 0;
    return r;
}
// R mod p (Montgomery one), R = 2^261
FR_HD __forceinline__ Fr fr_one() {
    Fr r = {{0x8fffff57u, 0x2fd4e156u, 0xa494b01au, 0x75bba827u, 0x819caa80u, 0x5301fa84u, 0x563d4475u, 0x0dc83629u}};
    return r;
}
FR_HD __forceinline__ bool fr_is_zero(const Fr &a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.v[i];
    return o == 0;
}
FR_HD __forceinline__ bool fr_eq(const Fr &a, const Fr &b) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.v[i] ^ b.v[i];
    return o == 0;
}

// 256-bit add / subtract as 8-limb carry chains (v_add_co_u32 + 7 x v_addc_co_u32 on gfx950)
FR_HD __forceinline__ uint32_t fr_add256(Fr &r, const Fr &a, const Fr &b) {
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = __builtin_addc(a.v[i], b.v[i], c, &c);
    return c;
}
FR_HD __forceinline__ uint32_t fr_sub256(Fr &r, const Fr &a, const Fr &b) {
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = __builtin_subc(a.v[i], b.v[i], c, &c);
    return c;
}
FR_HD __forceinline__ Fr fr_modulus() {
    Fr p = {{FR_P0, FR_P1, FR_P2, FR_P3, FR_P4, FR_P5, FR_P6, FR_P7}};
    return p;
}

// r = t - p if t >= p else t   (t < 2p)
FR_HD __forceinline__ Fr fr_cond_sub_p(const Fr &t) {
    Fr d;
    const uint32_t br = fr_sub256(d, t, fr_modulus());
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = br ? t.v[i] : d.v[i];
    return r;
}

FR_HD __forceinline__ Fr fr_add(const Fr &a, const Fr &b) {
    Fr t;
    fr_add256(t, a, b);  // a, b < p < 2^254: no carry out of 256 bits
    return fr_cond_sub_p(t);
}

FR_HD __forceinline__ Fr fr_sub(const Fr &a, const Fr &b) {
    Fr d, q;
    const uint32_t mask = fr_sub256(d, a, b) ? 0xffffffffu : 0u;
#pragma unroll
    for (int i = 0; i < 8; i++) q.v[i] = fr_p(i) & mask;
    Fr r;
    fr_add256(r, d, q);
    return r;
}

FR_HD __forceinline__ Fr fr_neg(const Fr &a) {
    Fr z = fr_zero();
    return fr_sub(z, a);  // 0 - 0 = 0; else p - a
}

// ---- working form: 9 x 29-bit limbs
struct Fr29 {
    uint32_t v[9];
};
FR_HD __forceinline__ uint32_t fr_p29(int i) {
    constexpr uint32_t P[9] = {0x10000001u, 0x1f0fac9fu, 0x0e5c2450u, 0x07d090f3u, 0x1585d283u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
    return P[i];
}
// 8 x 32 -> 9 x 29 (exact; limbs < 2^29, top limb < 2^24)
FR_HD __forceinline__ Fr29 fr29_from(const Fr &a) {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int bit = 29 * i, w = bit >> 5, sh = bit & 31;
        uint64_t two = a.v[w];
        if (w + 1 < 8) two |= (uint64_t)a.v[w + 1] << 32;
        r.v[i] = (uint32_t)(two >> sh) & 0x1fffffffu;
    }
    return r;
}
// 9 x 29 (limbs < 2^29, value < 2^256) -> 8 x 32
FR_HD __forceinline__ Fr fr29_pack(const Fr29 &a) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int bit = 32 * i, l = bit / 29, sh = bit - 29 * l;  // limb l holds bits [29 l, 29 l + 29)
        uint64_t acc = (uint64_t)a.v[l] >> sh;
        acc |= (uint64_t)a.v[l + 1] << (29 - sh);
        if (l + 2 < 9 && 58 - sh < 32) acc |= (uint64_t)a.v[l + 2] << (58 - sh);
        r.v[i] = (uint32_t)acc;
    }
    return r;
}
// Montgomery product a * b * 2^-261 mod p in the working form. Inputs: limbs < 2^29 (top limb < 2^28), values < 8p.
// Output: limbs < 2^29, value < 1.4p (< 1.06p for inputs < 4p). Column k of a * b + m * p is summed in one 64-bit
// accumulator (at most 18 products < 2^58 plus a 35-bit carry); m_k = column * (-p^-1) mod 2^29 zeroes the column's low
// limb, and -p^-1 = 2^28 - 1, p_0 = 2^28 + 1 turn both of those multiplications into shifts.
#ifdef FR_BLOCKS_ALL  // measurement only (tools/build_variant.sh): every fr29_mul of the translation unit takes the asm-block form below
#define fr29_mul fr29_mul_c
#endif
FR_HD __forceinline__ Fr29 fr29_mul(const Fr29 &a, const Fr29 &b) {
    constexpr uint32_t M = 0x1fffffffu;
    uint64_t acc = 0;
    uint32_t m[9];
    Fr29 r;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * fr_p29(k - i);
        const uint32_t lo = (uint32_t)acc;
        m[k] = (((lo & 1u) << 28) - lo) & M;
        acc += ((uint64_t)m[k] << 28) + m[k];
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i < 9; i++) acc += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
        for (int i = k - 8; i < 9; i++) acc += (uint64_t)m[i] * fr_p29(k - i);
        r.v[k - 9] = (uint32_t)acc & M;
        acc >>= 29;
    }
    r.v[8] = (uint32_t)acc;
    return r;
}
// Low 29 bits of the canonical value of a fully reduced Montgomery representative: a * 2^-261 mod p = (a + m p) / 2^261 is
// already < p for a < p (it can only reach p for a = 0, which gives 0), so its low limb is column 9 of the reduction and the
// upper seven columns are never formed: 44 multiply-adds instead of 72 + conditional subtraction + repack. For the byte-sized
// inputs of the hash black boxes (fetch_nearest_bytes of an 8-bit witness, generic_ark.rs:305-317).
FR_HD __forceinline__ uint32_t fr29_redc_low(const Fr29 &a) {
    constexpr uint32_t M = 0x1fffffffu;
    uint64_t acc = 0;
    uint32_t m[9];
#pragma unroll
    for (int k = 0; k < 9; k++) {
        acc += a.v[k];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * fr_p29(k - i);
        const uint32_t lo = (uint32_t)acc;
        m[k] = (((lo & 1u) << 28) - lo) & M;
        acc += ((uint64_t)m[k] << 28) + m[k];
        acc >>= 29;
    }
#pragma unroll
    for (int i = 1; i < 9; i++) acc += (uint64_t)m[i] * fr_p29(9 - i);
    return (uint32_t)acc & M;
}
// a * a * 2^-261 mod p: the 36 cross products are taken once against the doubled limbs (45 + 81 multiply-adds instead of
// 81 + 81). Same contract as fr29_mul.
FR_HD __forceinline__ Fr29 fr29_sqr(const Fr29 &a) {
    constexpr uint32_t M = 0x1fffffffu;
    uint32_t d[9];
#pragma unroll
    for (int i = 0; i < 9; i++) d[i] = a.v[i] << 1;
    uint64_t acc = 0;
    uint32_t m[9];
    Fr29 r;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; 2 * i < k; i++) acc += (uint64_t)d[i] * a.v[k - i];
        if ((k & 1) == 0) acc += (uint64_t)a.v[k / 2] * a.v[k / 2];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * fr_p29(k - i);
        const uint32_t lo = (uint32_t)acc;
        m[k] = (((lo & 1u) << 28) - lo) & M;
        acc += ((uint64_t)m[k] << 28) + m[k];
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; 2 * i < k; i++) acc += (uint64_t)d[i] * a.v[k - i];
        if ((k & 1) == 0) acc += (uint64_t)a.v[k / 2] * a.v[k / 2];
#pragma unroll
        for (int i = k - 8; i < 9; i++) acc += (uint64_t)m[i] * fr_p29(k - i);
        r.v[k - 9] = (uint32_t)acc & M;
        acc >>= 29;
    }
    r.v[8] = (uint32_t)acc;
    return r;
}
// value < 2p, limbs < 2^29 -> canonical [0, p)
FR_HD __forceinline__ Fr29 fr29_cond_sub_p(const Fr29 &a) {
    Fr29 d;
    int32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int32_t t = (int32_t)a.v[i] - (int32_t)fr_p29(i) + borrow;
        d.v[i] = (uint32_t)t & 0x1fffffffu;
        borrow = t >> 29;  // 0 or -1
    }
    Fr29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = borrow ? a.v[i] : d.v[i];
    return r;
}

// ---- lazy arithmetic on the working form (the Grumpkin point formulas): values are non-negative integers below a small
// multiple of p, limbs may exceed 29 bits between fr29_norm calls. A product needs normalised limbs and values < 16p.
// k * p with every limb below the top raised by 2^29 (and the borrow taken from the next one), so that limb-wise
// a + kp - b never goes negative below the top limb for a normalised b; the top limb may wrap, which cancels in fr29_norm.
FR_HD __forceinline__ uint32_t fr_kp29_sub(int klog2, int i) {
    constexpr uint32_t C[4][9] = {
        {0x20000002u, 0x3e1f593eu, 0x3cb848a0u, 0x2fa121e5u, 0x2b0ba505u, 0x25b68180u, 0x214dc281u, 0x3cb84c67u, 0x0060c89bu},   // 2p
        {0x20000004u, 0x3c3eb27du, 0x39709142u, 0x3f4243ccu, 0x36174a0bu, 0x2b6d0301u, 0x229b8503u, 0x397098cfu, 0x00c19138u},   // 4p
        {0x20000008u, 0x387d64fbu, 0x32e12286u, 0x3e84879au, 0x2c2e9418u, 0x36da0604u, 0x25370a07u, 0x32e1319fu, 0x01832272u},   // 8p
        {0x20000010u, 0x30fac9f7u, 0x25c2450eu, 0x3d090f36u, 0x385d2832u, 0x2db40c09u, 0x2a6e1410u, 0x25c2633fu, 0x030644e6u}};  // 16p
    return C[klog2 - 1][i];
}
FR_HD __forceinline__ uint32_t fr_kp29(int klog2, int i) {  // k * p, normalised limbs
    constexpr uint32_t C[4][9] = {
        {0x00000002u, 0x1e1f593fu, 0x1cb848a1u, 0x0fa121e6u, 0x0b0ba506u, 0x05b68181u, 0x014dc282u, 0x1cb84c68u, 0x0060c89cu},
        {0x00000004u, 0x1c3eb27eu, 0x19709143u, 0x1f4243cdu, 0x16174a0cu, 0x0b6d0302u, 0x029b8504u, 0x197098d0u, 0x00c19139u},
        {0x00000008u, 0x187d64fcu, 0x12e12287u, 0x1e84879bu, 0x0c2e9419u, 0x16da0605u, 0x05370a08u, 0x12e131a0u, 0x01832273u},
        {0x00000010u, 0x10fac9f8u, 0x05c2450fu, 0x1d090f37u, 0x185d2833u, 0x0db40c0au, 0x0a6e1411u, 0x05c26340u, 0x030644e7u}};
    return C[klog2 - 1][i];
}
FR_HD __forceinline__ Fr29 fr29_norm(const Fr29 &a) {  // carry propagation: limbs < 2^29 below the top one
    Fr29 r = a;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        r.v[i + 1] += r.v[i] >> 29;
        r.v[i] &= 0x1fffffffu;
    }
    return r;
}
FR_HD __forceinline__ Fr29 fr29_addl(const Fr29 &a, const Fr29 &b) {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + b.v[i];
    return r;
}
FR_HD __forceinline__ Fr29 fr29_dbll(const Fr29 &a) {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i] << 1;
    return r;
}
// a + (2^klog2) p - b; b must be normalised and value(b) <= 2^klog2 p
FR_HD __forceinline__ Fr29 fr29_subl(const Fr29 &a, const Fr29 &b, int klog2) {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + fr_kp29_sub(klog2, i) - b.v[i];
    return r;
}
// normalised a: a - (2^klog2) p if that is non-negative, else a
FR_HD __forceinline__ Fr29 fr29_csub(const Fr29 &a, int klog2) {
    Fr29 d;
    int32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int32_t t = (int32_t)a.v[i] - (int32_t)(klog2 ? fr_kp29(klog2, i) : fr_p29(i)) + borrow;
        d.v[i] = i < 8 ? ((uint32_t)t & 0x1fffffffu) : (uint32_t)t;
        borrow = i < 8 ? (t >> 29) : (t >> 31);
    }
    Fr29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = borrow ? a.v[i] : d.v[i];
    return r;
}
// normalised value < 8p -> < 2p
FR_HD __forceinline__ Fr29 fr29_lt2p(const Fr29 &a) { return fr29_csub(fr29_csub(a, 2), 1); }
// normalised limbs, ANY value below 2^261 (169 p) -> the same residue below 1.03 p, normalised, in 38 instructions (the two conditional
// subtractions above: 90). q = floor(floor(v / 2^248) floor(2^268 / p) / 2^20) is floor(v / p) or one less -- every rounding goes down, all of
// them together by less than 1.03 (2 000 000 values, the multiples of p and their neighbours among them: tools/fr_device_host_test.hip checks the
// contract) -- and v - q p = v + q (2^261 - p) - q 2^261 is summed limb by limb without a borrow.
FR_HD __forceinline__ Fr29 fr29_weak(const Fr29 &a) {
    constexpr uint32_t PP[9] = {0x0fffffffu, 0x00f05360u, 0x11a3dbafu, 0x182f6f0cu, 0x0a7a2d7cu, 0x1d24bf3fu, 0x1f591ebeu, 0x11a3d9cbu, 0x1fcf9bb1u};  // 2^261 - p
    const uint32_t q = ((a.v[8] >> 16) * 21668u) >> 20;  // < 170
    Fr29 r;
    uint64_t acc = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        acc += (uint64_t)q * PP[i] + a.v[i];
        r.v[i] = (uint32_t)acc & 0x1fffffffu;
        acc >>= 29;
    }
    acc += (uint64_t)q * PP[8] + a.v[8];
    r.v[8] = (uint32_t)(acc - ((uint64_t)q << 29));
    return r;
}
// normalised value < 8p -> canonical [0, p)
FR_HD __forceinline__ Fr29 fr29_canon(const Fr29 &a) { return fr29_csub(fr29_lt2p(a), 0); }
// normalised value < 2p: is it 0 mod p
FR_HD __forceinline__ bool fr29_is_zero_mod_p(const Fr29 &a) {
    uint32_t z = 0, e = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        z |= a.v[i];
        e |= a.v[i] ^ fr_p29(i);
    }
    return z == 0u || e == 0u;
}

// ---- sums of up to six products with ONE reduction: (sum_t a_t * b_t) * 2^-261 mod p. The arithmetic gate sum
// q_m a b + sum q_i w_i + q_c (pwg/arithmetic.rs:27-127) pays one Montgomery reduction (half of a product's 162
// multiply-adds) per group of terms instead of one per term. Same column scan as fr29_mul: column k of sum a_t b_t + m p is
// summed in one 64-bit accumulator ((N + 1) * 9 products < 2^58 plus a carry: N <= 6 fits). Contract: normalised limbs (below
// 2^29, the top one included); the VALUES only enter through the result's bound: result < p + (sum_t a_t b_t) / 2^261, i.e.
// 1 + 0.005908 sum_t (a_t / p)(b_t / p) in units of p -- below 1.04 p for two products of operands below 4 p, 1.34 p for two products
// of any two rows of the witness table (representatives below 2^256 = 5.29 p, gate_eval.hpp). Output limbs are normalised; the
// top limb carries whatever is left (value / 2^232).
// ADD: a lazy sum h (limbs below 2^32, not normalised) rides in the upper columns: result = (sum a_t b_t + m p) / 2^261 + h exactly,
// with the carries of h propagated by the scan (the gate kernel's constant and +-1 terms: no separate carry pass).
template <int N, bool ADD>
FR_HD __forceinline__ Fr29 fr29_dot_impl(const Fr29 (&a)[N], const Fr29 (&b)[N], const Fr29 *h) {
    static_assert(N >= 1 && N <= 6, "column accumulator budget");
    constexpr uint32_t M = 0x1fffffffu;
    uint64_t acc = 0;
    uint32_t m[9];
    Fr29 r;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int t = 0; t < N; t++)
#pragma unroll
            for (int i = 0; i <= k; i++) acc += (uint64_t)a[t].v[i] * b[t].v[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * fr_p29(k - i);
        const uint32_t lo = (uint32_t)acc;
        m[k] = (((lo & 1u) << 28) - lo) & M;
        acc += ((uint64_t)m[k] << 28) + m[k];
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int t = 0; t < N; t++)
#pragma unroll
            for (int i = k - 8; i < 9; i++) acc += (uint64_t)a[t].v[i] * b[t].v[k - i];
#pragma unroll
        for (int i = k - 8; i < 9; i++) acc += (uint64_t)m[i] * fr_p29(k - i);
        if (ADD) acc += h->v[k - 9];
        r.v[k - 9] = (uint32_t)acc & M;
        acc >>= 29;
    }
    r.v[8] = (uint32_t)acc;
    if (ADD) r.v[8] += h->v[8];
    return r;
}
template <int N>
FR_HD __forceinline__ Fr29 fr29_dot(const Fr29 (&a)[N], const Fr29 (&b)[N]) { return fr29_dot_impl<N, false>(a, b, nullptr); }
template <int N>
FR_HD __forceinline__ Fr29 fr29_dot_add(const Fr29 (&a)[N], const Fr29 (&b)[N], const Fr29 &h) { return fr29_dot_impl<N, true>(a, b, &h); }

// ---- the same scans as asm blocks (device pass; generated: tools/gen_mul_blocks.py > fr_blocks.inc, which says why). hipcc's form of the scan above
// joins every column with the carry of the one before by a 64-bit add -- 17 of a product's 221 VALU instructions; a column whose first multiply-add
// takes the carry needs none. For kernels that keep four or more waves per SIMD (the gate kernels, gate_eval.hpp): at two waves per SIMD the serial
// chain gains nothing (tools/serial_mul_probe.hip), so the Grumpkin / ECDSA kernels stay with the compiler's form. Same values, same contracts; the
// host pass (and -DFR_NO_ASM_BLOCKS) takes the C forms. UB: bit t set = every limb of b[t] is WAVE-UNIFORM (a gate's coefficient: scalar registers).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(FR_NO_ASM_BLOCKS)
#define FR_ASM_BLOCKS 1
#include "fr_blocks.inc"
#endif
#ifdef FR_BLOCKS_ALL
#undef fr29_mul
FR_HD __forceinline__ Fr29 fr29_mul(const Fr29 &a, const Fr29 &b) {
#if FR_ASM_BLOCKS
    return fr29_mul_blk(a, b);
#else
    return fr29_mul_c(a, b);
#endif
}
#endif
FR_HD __forceinline__ Fr29 fr29_mul_b(const Fr29 &a, const Fr29 &b) {
#if FR_ASM_BLOCKS
    return fr29_mul_blk(a, b);
#else
    return fr29_mul(a, b);
#endif
}
template <int N, unsigned UB>
FR_HD __forceinline__ Fr29 fr29_dot_add_b(const Fr29 (&a)[N], const Fr29 (&b)[N], const Fr29 &h) {
    static_assert(N == 1 || N == 2, "generated forms");
#if FR_ASM_BLOCKS
    if constexpr (N == 1) {
        static_assert(UB <= 1u, "one product");
        if constexpr (UB == 1u) return fr29_dot1_add_blk_u(a[0], b[0], h);
        else return fr29_dot1_add_blk_v(a[0], b[0], h);
    } else {
        static_assert(UB == 0u || UB == 2u || UB == 3u, "a uniform first factor pairs with a uniform second one");
        if constexpr (UB == 0u) return fr29_dot2_add_blk_vv(a[0], b[0], a[1], b[1], h);
        else if constexpr (UB == 2u) return fr29_dot2_add_blk_vu(a[0], b[0], a[1], b[1], h);
        else return fr29_dot2_add_blk_uu(a[0], b[0], a[1], b[1], h);
    }
#else
    return fr29_dot_add<N>(a, b, h);
#endif
}

// Montgomery product on the storage form, fully reduced
FR_HD __forceinline__ Fr fr_mul(const Fr &a, const Fr &b) { return fr29_pack(fr29_cond_sub_p(fr29_mul(fr29_from(a), fr29_from(b)))); }

// Independent reference for the self tests: the textbook 8 x 32-limb CIOS with R = 2^256, applied twice
// (a * b * 2^-256, then * 2^251 * 2^-256 = * 2^-5) to land in the same R = 2^261 domain.
FR_HD __forceinline__ Fr fr_cios256(const Fr &a, const Fr &b) {
    uint32_t t[8];
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t A = (uint64_t)a.v[0] * b.v[i] + t[0];
        uint32_t m = (uint32_t)A * FR_N0INV;
        uint64_t C = (uint64_t)m * FR_P0 + (uint32_t)A;
        A >>= 32;
        C >>= 32;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            A += (uint64_t)a.v[j] * b.v[i] + t[j];
            C += (uint64_t)m * fr_p(j) + (uint32_t)A;
            t[j - 1] = (uint32_t)C;
            A >>= 32;
            C >>= 32;
        }
        t[7] = (uint32_t)(C + A);
    }
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = t[i];
    return fr_cond_sub_p(r);
}
FR_HD __forceinline__ Fr fr_mul_portable(const Fr &a, const Fr &b) {
    Fr two251 = fr_zero();
    two251.v[7] = 1u << 27;  // 2^251 < p
    return fr_cios256(fr_cios256(a, b), two251);
}

FR_HD __forceinline__ Fr fr_sqr(const Fr &a) { return fr_mul(a, a); }

// R^3 mod p: takes the integer inverse of a Montgomery representative back into Montgomery form
FR_HD __forceinline__ Fr fr_r3() {
    Fr r = {{0x601fddb2u, 0xbafa616cu, 0x23e89803u, 0x29b4a83eu, 0x44d1496bu, 0x917ad601u, 0xfe59ed6du, 0x1baa96fcu}};
    return r;
}

// Reference implementation kept for the device self test (acvm_selftest cross-checks it against fr_inv): binary extended Euclid on the Montgomery
// representative x = aR (an integer < p) with invariants b*x == u, c*x == v (mod p), written branch-free so
// that the 64 lanes of a wavefront stay converged: every iteration halves u (after subtracting the smaller
// odd value when u is odd; the pairs are swapped by selects so that u is the larger one), ~120 VALU ops per
// iteration, at most ~2*254 iterations. When u reaches 0, v = gcd = 1 and c = x^-1 = a^-1 R^-1; one
// Montgomery product with R^3 returns a^-1 R. For a == 0 the loop never runs and c == 0.
FR_HD inline __noinline__ Fr fr_inv_eea(const Fr &a) {
    Fr u = a, v = fr_modulus(), b = fr_zero(), c = fr_zero();
    b.v[0] = 1;
    while (!fr_is_zero(u)) {
        const bool odd = u.v[0] & 1u;
        Fr t;
        const bool swap = odd && fr_sub256(t, u, v);  // u odd and u < v
        Fr x, y, bx, by;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            x.v[i] = swap ? v.v[i] : u.v[i];
            y.v[i] = swap ? u.v[i] : v.v[i];
            bx.v[i] = swap ? c.v[i] : b.v[i];
            by.v[i] = swap ? b.v[i] : c.v[i];
        }
        const uint32_t om = odd ? 0xffffffffu : 0u;
        Fr ym, bym;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            ym.v[i] = y.v[i] & om;
            bym.v[i] = by.v[i] & om;
        }
        fr_sub256(x, x, ym);  // x >= y when odd: no borrow; now x is even
#pragma unroll
        for (int i = 0; i < 7; i++) x.v[i] = x.v[i] >> 1 | x.v[i + 1] << 31;
        x.v[7] >>= 1;
        // bx = (bx - bym) / 2 mod p
        const uint32_t neg = fr_sub256(bx, bx, bym) ? 0xffffffffu : 0u;
        Fr q;
#pragma unroll
        for (int i = 0; i < 8; i++) q.v[i] = fr_p(i) & neg;
        fr_add256(bx, bx, q);
        const uint32_t bodd = (bx.v[0] & 1u) ? 0xffffffffu : 0u;
#pragma unroll
        for (int i = 0; i < 8; i++) q.v[i] = fr_p(i) & bodd;
        const uint32_t top = fr_add256(bx, bx, q);  // bx + p < 2p < 2^255: top is always 0, kept for clarity
#pragma unroll
        for (int i = 0; i < 7; i++) bx.v[i] = bx.v[i] >> 1 | bx.v[i + 1] << 31;
        bx.v[7] = bx.v[7] >> 1 | top << 31;
        u = x;
        v = y;
        b = bx;
        c = by;
    }
    return fr_mul(c, fr_r3());
}

// Field inverse, inverse(0) == 0 (generic_ark.rs:242-245): Bernstein-Yang "safegcd" with the half-delta divsteps in
// the constant-time form (the schedule of libsecp256k1's modinv32: 20 batches of 30 divsteps cover any 256-bit input).
// Every lane runs the identical instruction stream -- no data-dependent trip count, so a wavefront never diverges and an
// inversion costs ~19k issue slots instead of the ~75k of the binary Euclid above.
//   f = p, g = x (the Montgomery representative aR as an integer), d = 0, e = 1 with d*x == f, e*x == g (mod p);
//   30 divsteps on the low words give a 2x2 transition matrix t (entries < 2^30 in magnitude);
//   [f, g] <- t [f, g] / 2^30 exactly, [d, e] <- t [d, e] / 2^30 mod p (a multiple of p cancels the low 30 bits).
// Numbers are 9 signed limbs of 30 bits. After 600 divsteps g == 0, f == +-1 and d == +-x^-1.
struct FrS30 {
    int32_t v[9];
};
// a * b + c with a, b signed 32-bit: one v_mad_i64_i32 (hipcc otherwise widens the product to a 64 x 64 multiplication)
FR_HD __forceinline__ int64_t fr_mad_i64(int32_t a, int32_t b, int64_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    int64_t r;
    asm("v_mad_i64_i32 %0, vcc, %1, %2, %3" : "=&v"(r) : "v"(a), "v"(b), "v"(c) : "vcc");
    return r;
#else
    return (int64_t)a * b + c;
#endif
}
// The modulus of a safegcd inversion: its 9 x 30-bit limbs and its inverse modulo 2^30. BN254-Fr here; ops_ecdsa.hpp instantiates the same
// routine for the four moduli of secp256k1 / secp256r1 (any odd modulus below 2^256 works: 600 divsteps cover 256-bit inputs).
struct FrMod30 {
    static FR_HD __forceinline__ int32_t p30(int i) {
        constexpr int32_t P30[9] = {0x30000001, 0x0f87d64f, 0x1b970914, 0x0cfa121e, 0x01585d28, 0x0116da06, 0x1a029b85, 0x139cb84c, 0x3064};
        return P30[i];
    }
    static constexpr uint32_t PINV30 = 0x10000001u;  // p^-1 mod 2^30
};
// x^-1 mod M for the integer x < M held in a (0 for x = 0), as an integer < M
template <class M>
FR_HD __forceinline__ Fr fr_safegcd_inv(const Fr &a) {
    constexpr int32_t M30 = 0x3fffffff;
    constexpr uint32_t PINV30 = M::PINV30;
    FrS30 d, e, f, g;
    // 8 x 32 -> 9 x 30
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int bit = 30 * i, w = bit >> 5, sh = bit & 31;
        uint64_t two = a.v[w];
        if (w + 1 < 8) two |= (uint64_t)a.v[w + 1] << 32;
        g.v[i] = (int32_t)((uint32_t)(two >> sh) & (uint32_t)M30);
        f.v[i] = M::p30(i);
        d.v[i] = 0;
        e.v[i] = i == 0 ? 1 : 0;
    }
    int32_t zeta = -1;  // -(delta + 1/2), delta = 1/2
    for (int it = 0; it < 20; it++) {
        // ---- 30 divsteps on the low limbs
        uint32_t u = 1, v = 0, q = 0, r = 1, ff = (uint32_t)f.v[0], gg = (uint32_t)g.v[0];
#pragma unroll 6
        for (int i = 0; i < 30; i++) {
            uint32_t m1 = (uint32_t)(zeta >> 31);  // zeta < 0
            const uint32_t m2 = 0u - (gg & 1u);    // g odd
            const uint32_t x = (ff ^ m1) - m1, y = (u ^ m1) - m1, z = (v ^ m1) - m1;
            gg += x & m2;
            q += y & m2;
            r += z & m2;
            m1 &= m2;
            zeta = (int32_t)((uint32_t)zeta ^ m1) - 1;
            ff += gg & m1;
            u += q & m1;
            v += r & m1;
            gg >>= 1;
            u <<= 1;
            v <<= 1;
        }
        const int32_t tu = (int32_t)u, tv = (int32_t)v, tq = (int32_t)q, tr = (int32_t)r;
        // ---- [d, e] <- t [d, e] / 2^30 mod p
        {
            const int32_t sd = d.v[8] >> 31, se = e.v[8] >> 31;
            int32_t md = (tu & sd) + (tv & se), me = (tq & sd) + (tr & se);
            int64_t cd = fr_mad_i64(tu, d.v[0], fr_mad_i64(tv, e.v[0], 0));
            int64_t ce = fr_mad_i64(tq, d.v[0], fr_mad_i64(tr, e.v[0], 0));
            md -= (int32_t)((PINV30 * (uint32_t)cd + (uint32_t)md) & (uint32_t)M30);
            me -= (int32_t)((PINV30 * (uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
            cd = fr_mad_i64(M::p30(0), md, cd);
            ce = fr_mad_i64(M::p30(0), me, ce);
            cd >>= 30;
            ce >>= 30;
#pragma unroll
            for (int i = 1; i < 9; i++) {
                const int32_t di = d.v[i], ei = e.v[i];
                cd = fr_mad_i64(tu, di, fr_mad_i64(tv, ei, fr_mad_i64(M::p30(i), md, cd)));
                ce = fr_mad_i64(tq, di, fr_mad_i64(tr, ei, fr_mad_i64(M::p30(i), me, ce)));
                d.v[i - 1] = (int32_t)cd & M30;
                e.v[i - 1] = (int32_t)ce & M30;
                cd >>= 30;
                ce >>= 30;
            }
            d.v[8] = (int32_t)cd;
            e.v[8] = (int32_t)ce;
        }
        // ---- [f, g] <- t [f, g] / 2^30 (exact)
        {
            int64_t cf = fr_mad_i64(tu, f.v[0], fr_mad_i64(tv, g.v[0], 0));
            int64_t cg = fr_mad_i64(tq, f.v[0], fr_mad_i64(tr, g.v[0], 0));
            cf >>= 30;
            cg >>= 30;
#pragma unroll
            for (int i = 1; i < 9; i++) {
                const int32_t fi = f.v[i], gi = g.v[i];
                cf = fr_mad_i64(tu, fi, fr_mad_i64(tv, gi, cf));
                cg = fr_mad_i64(tq, fi, fr_mad_i64(tr, gi, cg));
                f.v[i - 1] = (int32_t)cf & M30;
                g.v[i - 1] = (int32_t)cg & M30;
                cf >>= 30;
                cg >>= 30;
            }
            f.v[8] = (int32_t)cf;
            g.v[8] = (int32_t)cg;
        }
    }
    // normalise d * sign(f) from (-2p, p) to [0, p)
    {
        int32_t cond_add = d.v[8] >> 31;
        const int32_t cond_neg = f.v[8] >> 31;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            d.v[i] += M::p30(i) & cond_add;
            d.v[i] = (d.v[i] ^ cond_neg) - cond_neg;
        }
#pragma unroll
        for (int i = 0; i < 8; i++) { d.v[i + 1] += d.v[i] >> 30; d.v[i] &= M30; }
        cond_add = d.v[8] >> 31;
#pragma unroll
        for (int i = 0; i < 9; i++) d.v[i] += M::p30(i) & cond_add;
#pragma unroll
        for (int i = 0; i < 8; i++) { d.v[i + 1] += d.v[i] >> 30; d.v[i] &= M30; }
    }
    // 9 x 30 -> 8 x 32
    Fr c;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int bit = 32 * i, l = bit / 30, sh = bit - 30 * l;  // limb l holds bits [30 l, 30 l + 30)
        uint64_t acc = (uint64_t)(uint32_t)d.v[l] >> sh;
        acc |= (uint64_t)(uint32_t)d.v[l + 1] << (30 - sh);
        if (l + 2 < 9) acc |= (uint64_t)(uint32_t)d.v[l + 2] << (60 - sh);
        c.v[i] = (uint32_t)acc;
    }
    return c;
}
// x^-1 = a^-1 R^-1 as an integer < p for the Montgomery representative x = a R; one Montgomery product with R^3 returns a^-1 R
FR_HD inline __noinline__ Fr fr_inv(const Fr &a) { return fr_mul(fr_safegcd_inv<FrMod30>(a), fr_r3()); }

// ---- witness table access: W[slot][half][instance] of 16-byte units (limbs 0..3 / 4..7), so that a
// wavefront's 64 lanes read 1 KiB contiguous per instruction.
FR_HD __forceinline__ Fr fr_load(const uint4 *W, uint32_t slot, uint64_t B, uint64_t j) {
    const uint4 lo = W[(uint64_t)slot * 2 * B + j];
    const uint4 hi = W[((uint64_t)slot * 2 + 1) * B + j];
    Fr r = {{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w}};
    return r;
}
FR_HD __forceinline__ void fr_store(uint4 *W, uint32_t slot, uint64_t B, uint64_t j, const Fr &a) {
    W[(uint64_t)slot * 2 * B + j] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
    W[((uint64_t)slot * 2 + 1) * B + j] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
}
// the same store marked nontemporal (global_store ... nt): a row that no wave will read again before it has left the caches anyway
// (outputs of the hash kernels: config 3 0.447 -> 0.505 of the HBM roofline) does not displace lines that are still to be read.
// One 16-byte access per half, spelled as a vector so that the compiler cannot split a half into overlapping dwordx3 + dwordx2 loads
// (it did, once the eight scalar loads were inlined through a loader object: 3 VMEM instructions per row instead of 2).
typedef uint32_t fr_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void fr_store_nt(uint4 *W, uint32_t slot, uint64_t B, uint64_t j, const Fr &a) {
    uint4 *lo = W + (uint64_t)slot * 2 * B + j, *hi = lo + B;
    const fr_u32x4 l = {a.v[0], a.v[1], a.v[2], a.v[3]}, h = {a.v[4], a.v[5], a.v[6], a.v[7]};
    __builtin_nontemporal_store(l, (fr_u32x4 *)lo);
    __builtin_nontemporal_store(h, (fr_u32x4 *)hi);
}
__device__ __forceinline__ Fr fr_load_nt(const uint4 *W, uint32_t slot, uint64_t B, uint64_t j) {
    const uint4 *lo = W + (uint64_t)slot * 2 * B + j, *hi = lo + B;
    const fr_u32x4 l = __builtin_nontemporal_load((const fr_u32x4 *)lo), h = __builtin_nontemporal_load((const fr_u32x4 *)hi);
    Fr r = {{l.x, l.y, l.z, l.w, h.x, h.y, h.z, h.w}};
    return r;
}
// circuit constant (wave-uniform): 8 consecutive u32 in the constants table
FR_HD __forceinline__ Fr fr_const(const uint32_t *__restrict__ consts, uint32_t idx) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = consts[(uint64_t)idx * 8 + i];
    return r;
}

}  // namespace acvm
