// fr_device.hpp -- BN254-Fr arithmetic for gfx950 (CDNA4), the inner loop of every kernel.
// Replaces acir_field::FieldElement add/sub/neg/mul/inverse (acir_field/src/generic_ark.rs:242-245,
// 360-406; ark-ff Fp256<MontBackend<_,4>>). Representation: Montgomery form, R = 2^256, as 8 x 32-bit
// limbs held in VGPRs (CDNA4 has no 64x64 multiplier; the 32x32->64 v_mad_u64_u32 is the native wide
// multiply-add). Values are always fully reduced to [0, p), so equality / is_zero are limb compares,
// matching the reference's canonical-bytes equality (generic_ark.rs:88-92,164-169).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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

__device__ __forceinline__ uint32_t fr_p(int i) {
    constexpr uint32_t P[8] = {FR_P0, FR_P1, FR_P2, FR_P3, FR_P4, FR_P5, FR_P6, FR_P7};
    return P[i];
}

__device__ __forceinline__ Fr fr_zero() {
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = 0;
    return r;
}
// R mod p (Montgomery one)
__device__ __forceinline__ Fr fr_one() {
    Fr r = {{0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u, 0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u}};
    return r;
}
__device__ __forceinline__ bool fr_is_zero(const Fr &a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.v[i];
    return o == 0;
}
__device__ __forceinline__ bool fr_eq(const Fr &a, const Fr &b) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.v[i] ^ b.v[i];
    return o == 0;
}

// r = t - p if t >= p else t   (t < 2p)
__device__ __forceinline__ Fr fr_cond_sub_p(const Fr &t) {
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

__device__ __forceinline__ Fr fr_add(const Fr &a, const Fr &b) {
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

__device__ __forceinline__ Fr fr_sub(const Fr &a, const Fr &b) {
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

__device__ __forceinline__ Fr fr_neg(const Fr &a) {
    Fr z = fr_zero();
    return fr_sub(z, a);  // 0 - 0 = 0; else p - a
}

// Montgomery product a*b*R^-1 mod p, fully reduced. CIOS with the "no-carry" simplification that the
// spare top bits of p allow (p < 2^254): two 32x32+64 multiply-adds per limb pair.
__device__ __forceinline__ Fr fr_mul(const Fr &a, const Fr &b) {
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

__device__ __forceinline__ Fr fr_sqr(const Fr &a) { return fr_mul(a, a); }

// a^(p-2): inverse by Fermat (inverse(0) == 0 falls out: 0^(p-2) = 0). generic_ark.rs:242-245.
__device__ __noinline__ Fr fr_inv(const Fr &a) {
    // p - 2, little-endian 32-bit words
    const uint32_t E[8] = {FR_P0 - 2u, FR_P1, FR_P2, FR_P3, FR_P4, FR_P5, FR_P6, FR_P7};
    Fr r = fr_one();
    for (int w = 7; w >= 0; w--) {
        uint32_t e = E[w];
        for (int b = 31; b >= 0; b--) {
            r = fr_sqr(r);
            if ((e >> b) & 1) r = fr_mul(r, a);
        }
    }
    return r;
}

// ---- witness table access: W[slot][half][instance] of 16-byte units (limbs 0..3 / 4..7), so that a
// wavefront's 64 lanes read 1 KiB contiguous per instruction.
__device__ __forceinline__ Fr fr_load(const uint4 *W, uint32_t slot, uint64_t B, uint64_t j) {
    const uint4 lo = W[(uint64_t)slot * 2 * B + j];
    const uint4 hi = W[((uint64_t)slot * 2 + 1) * B + j];
    Fr r = {{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w}};
    return r;
}
__device__ __forceinline__ void fr_store(uint4 *W, uint32_t slot, uint64_t B, uint64_t j, const Fr &a) {
    W[(uint64_t)slot * 2 * B + j] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
    W[((uint64_t)slot * 2 + 1) * B + j] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
}
// circuit constant (wave-uniform): 8 consecutive u32 in the constants table
__device__ __forceinline__ Fr fr_const(const uint32_t *__restrict__ consts, uint32_t idx) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = consts[(uint64_t)idx * 8 + i];
    return r;
}

}  // namespace acvm
