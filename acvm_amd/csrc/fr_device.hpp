// fr_device.hpp -- BN254-Fr arithmetic for gfx950 (CDNA4), the inner loop of every kernel.
// Replaces acir_field::FieldElement add/sub/neg/mul/inverse (acir_field/src/generic_ark.rs:242-245,
// 360-406; ark-ff Fp256<MontBackend<_,4>>). Representation: Montgomery form, R = 2^256, as 8 x 32-bit
// limbs held in VGPRs (CDNA4 has no 64x64 multiplier; the 32x32->64 v_mad_u64_u32 is the native wide
// multiply-add). Values are always fully reduced to [0, p), so equality / is_zero are limb compares,
// matching the reference's canonical-bytes equality (generic_ark.rs:88-92,164-169).
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
// R mod p (Montgomery one)
FR_HD __forceinline__ Fr fr_one() {
    Fr r = {{0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u, 0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u}};
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

// r = t - p if t >= p else t   (t < 2p)
FR_HD __forceinline__ Fr fr_cond_sub_p(const Fr &t) {
    Fr d;
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t x = (uint64_t)t.v[i] - fr_p(i) - br;
        d.v[i] = (uint32_t)x;
        br = (x >> 32) & 1;
    }
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = br ? t.v[i] : d.v[i];
    return r;
}

FR_HD __forceinline__ Fr fr_add(const Fr &a, const Fr &b) {
    Fr t;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)a.v[i] + b.v[i];
        t.v[i] = (uint32_t)c;
        c >>= 32;
    }
    // a, b < p < 2^254: no carry out of 256 bits
    return fr_cond_sub_p(t);
}

FR_HD __forceinline__ Fr fr_sub(const Fr &a, const Fr &b) {
    Fr d;
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t x = (uint64_t)a.v[i] - b.v[i] - br;
        d.v[i] = (uint32_t)x;
        br = (x >> 32) & 1;
    }
    uint32_t mask = br ? 0xffffffffu : 0u;
    uint64_t c = 0;
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)d.v[i] + (fr_p(i) & mask);
        r.v[i] = (uint32_t)c;
        c >>= 32;
    }
    return r;
}

FR_HD __forceinline__ Fr fr_neg(const Fr &a) {
    Fr z = fr_zero();
    return fr_sub(z, a);  // 0 - 0 = 0; else p - a
}

// Montgomery product a*b*R^-1 mod p, fully reduced. CIOS with the "no-carry" simplification that the
// spare top bits of p allow (p < 2^254): two 32x32+64 multiply-adds per limb pair.
FR_HD __forceinline__ Fr fr_mul(const Fr &a, const Fr &b) {
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

FR_HD __forceinline__ Fr fr_sqr(const Fr &a) { return fr_mul(a, a); }

// R^3 mod p: takes the integer inverse of a Montgomery representative back into Montgomery form
FR_HD __forceinline__ Fr fr_r3() {
    Fr r = {{0xb4bf0040u, 0x5e94d8e1u, 0x1cfbb6b8u, 0x2a489cbeu, 0xa19fcfedu, 0x893cc664u, 0x7fcc657cu, 0x0cf8594bu}};
    return r;
}

// Field inverse, inverse(0) == 0 (generic_ark.rs:242-245). Binary extended Euclid with one halving per
// iteration on the Montgomery representative x = aR (an integer < p): invariants b*x == u, c*x == v (mod p).
// The result c = x^-1 = a^-1 R^-1 is multiplied by R^3 (one Montgomery product) to give a^-1 R.
// About 1.4 * 254 iterations of shifts/adds on 256-bit values: ~10x cheaper than a Fermat ladder on a
// machine whose 32x32 multiplier runs at quarter rate.
FR_HD __noinline__ Fr fr_inv(const Fr &a) {
    uint64_t u[4], v[4], b[4] = {1, 0, 0, 0}, c[4] = {0, 0, 0, 0};
    const uint64_t P[4] = {(uint64_t)FR_P1 << 32 | FR_P0, (uint64_t)FR_P3 << 32 | FR_P2, (uint64_t)FR_P5 << 32 | FR_P4,
                           (uint64_t)FR_P7 << 32 | FR_P6};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        u[i] = (uint64_t)a.v[2 * i + 1] << 32 | a.v[2 * i];
        v[i] = P[i];
    }
    auto is_zero = [](const uint64_t *x) { return (x[0] | x[1] | x[2] | x[3]) == 0; };
    auto shr1 = [](uint64_t *x, uint64_t top) {
        x[0] = x[0] >> 1 | x[1] << 63;
        x[1] = x[1] >> 1 | x[2] << 63;
        x[2] = x[2] >> 1 | x[3] << 63;
        x[3] = x[3] >> 1 | top << 63;
    };
    auto add4 = [](uint64_t *r, const uint64_t *x, const uint64_t *y) -> uint64_t {  // returns carry
        uint64_t cy = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            uint64_t s = x[i] + y[i], c1 = s < x[i];
            uint64_t s2 = s + cy, c2 = s2 < s;
            r[i] = s2;
            cy = c1 | c2;
        }
        return cy;
    };
    auto sub4 = [](uint64_t *r, const uint64_t *x, const uint64_t *y) -> uint64_t {  // returns borrow
        uint64_t br = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            uint64_t d = x[i] - y[i], b1 = x[i] < y[i];
            uint64_t d2 = d - br, b2 = d < br;
            r[i] = d2;
            br = b1 | b2;
        }
        return br;
    };
    auto halve_mod = [&](uint64_t *x) {  // x/2 mod p for x < p
        uint64_t cy = 0;
        if (x[0] & 1) cy = add4(x, x, P);
        shr1(x, cy);
    };
    auto sub_mod = [&](uint64_t *x, const uint64_t *y) {  // x - y mod p
        if (sub4(x, x, y)) add4(x, x, P);
    };
    while (!is_zero(u)) {
        if (!(u[0] & 1)) {
            shr1(u, 0);
            halve_mod(b);
        } else if (!(v[0] & 1)) {
            shr1(v, 0);
            halve_mod(c);
        } else {
            uint64_t t[4];
            if (!sub4(t, u, v)) {  // u >= v: u = (u - v) / 2
#pragma unroll
                for (int i = 0; i < 4; i++) u[i] = t[i];
                shr1(u, 0);
                sub_mod(b, c);
                halve_mod(b);
            } else {  // v = (v - u) / 2
                sub4(v, v, u);
                shr1(v, 0);
                sub_mod(c, b);
                halve_mod(c);
            }
        }
    }
    // a == 0: the loop never runs and c == 0; otherwise gcd = v = 1 and c = x^-1
    Fr r;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        r.v[2 * i] = (uint32_t)c[i];
        r.v[2 * i + 1] = (uint32_t)(c[i] >> 32);
    }
    return fr_mul(r, fr_r3());
}

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
// circuit constant (wave-uniform): 8 consecutive u32 in the constants table
FR_HD __forceinline__ Fr fr_const(const uint32_t *__restrict__ consts, uint32_t idx) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = consts[(uint64_t)idx * 8 + i];
    return r;
}

}  // namespace acvm
