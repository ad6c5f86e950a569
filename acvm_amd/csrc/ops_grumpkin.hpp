// ops_grumpkin.hpp -- device routines of the three BlackBoxFunctionSolver functions (blackbox_solver/src/lib.rs:27-45)
// that the reference delegates to barretenberg (barretenberg_blackbox_solver/src/{lib.rs:39-81, wasm/*.rs}); the
// algorithms are the ones SURVEY.md Appendix A specifies and the reference's five golden vectors pin:
//   FixedBaseScalarMul   acvm/src/pwg/blackbox/fixed_base_scalar_mul.rs:11-27, wasm/scalar_mul.rs:17-65
//   Pedersen             acvm/src/pwg/blackbox/pedersen.rs:11-28, wasm/pedersen.rs:14-35 (plookup commitment)
//   SchnorrVerify        acvm/src/pwg/blackbox/signature/schnorr.rs:13-35, lib.rs:40-58, wasm/schnorr.rs:68-103
// Grumpkin is y^2 = x^3 - 17 over BN254-Fr, so point coordinates are this library's Fr; scalars are 256-bit integers
// (8 x u32, canonical). One lane = one instance; table lookups are per-lane gathers from L2-resident tables
// (grumpkin_host.cpp). Integer-ALU bound: ~700 field multiplications per Pedersen hash_pair, ~5000 per Schnorr verify.
#pragma once
#include "grumpkin_host.hpp"
#include "ops_hash.hpp"

namespace acvm {

// Affine points travel in the storage form (tables, witnesses); Jacobian accumulators live in the 29-bit working form with
// LAZY reduction (fr_device.hpp): coordinates are kept "class A" = normalised limbs, value < 2p, and inside a formula sums
// and differences are limb-wise (a difference adds a multiple of p first) and only renormalised before they feed a product,
// which accepts values < 16p and returns < 1.4p. The bounds are written beside each line (in units of p). This removes every
// pack / unpack / conditional subtraction between the 11-16 products of a point operation (about 1.3x fewer instructions).
struct GAff { Fr x, y; };
struct GJac { Fr29 X, Y, Z; };  // class A coordinates; Z == 0 (mod p) <=> point at infinity

__device__ __forceinline__ Fr29 g29_one() { return fr29_from(fr_one()); }
__device__ __forceinline__ Fr29 g29_zero() {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = 0;
    return r;
}
__device__ __forceinline__ GJac gj_inf() { return GJac{g29_one(), g29_one(), g29_zero()}; }
__device__ __forceinline__ bool gj_is_inf(const GJac &p) { return fr29_is_zero_mod_p(p.Z); }
__device__ __forceinline__ Fr29 g29_red(const Fr29 &loose) { return fr29_weak(fr29_norm(loose)); }  // loose value < 169p -> class A (below 1.03p)

__device__ __forceinline__ GAff gaff_load(const uint4 *tbl, uint32_t idx) {
    const uint4 *p = tbl + (uint64_t)idx * 4;
    const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
    GAff r;
    r.x = Fr{{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
    r.y = Fr{{c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w}};
    return r;
}

// dbl-2009-l (a = 0): 2M + 5S
__device__ __forceinline__ GJac gj_dbl(const GJac &p) {
    if (gj_is_inf(p) || fr29_is_zero_mod_p(p.Y)) return gj_inf();
    const Fr29 A = fr29_sqr(p.X), B = fr29_sqr(p.Y), C = fr29_sqr(B);           // < 1.03, 1.03, 1.01
    const Fr29 t0 = fr29_norm(fr29_addl(p.X, B));                                              // < 3.03
    const Fr29 t = g29_red(fr29_subl(fr29_subl(fr29_sqr(t0), A, 1), C, 1));                // 1.06 + 4 = 5.06 -> < 2
    const Fr29 D = fr29_csub(fr29_norm(fr29_dbll(t)), 1);                                      // 2t < 4 -> < 2
    const Fr29 E = fr29_norm(fr29_addl(fr29_dbll(A), A));                                      // 3A < 3.09
    const Fr29 F = fr29_sqr(E);                                                             // < 1.06
    GJac r;
    r.X = g29_red(fr29_subl(F, fr29_norm(fr29_dbll(D)), 2));                                   // 1.06 + 4 = 5.06 -> < 2
    const Fr29 m = fr29_mul(E, fr29_norm(fr29_subl(D, r.X, 1)));                               // E (< 3.09) * (< 4) -> < 1.08
    const Fr29 C4 = g29_red(fr29_dbll(fr29_dbll(C)));                                          // 4C < 4.04 -> < 2
    r.Y = g29_red(fr29_subl(m, fr29_norm(fr29_dbll(C4)), 2));                                  // 1.08 + 4 = 5.08 -> < 2
    r.Z = fr29_mul(fr29_norm(fr29_dbll(p.Y)), p.Z);                                            // (2 Y) Z: 1 + 4 * 2 / 169 -> < 1.05 (the doubling before the product: no reduction behind it)
    return r;
}
// complete mixed addition (madd-2007-bl with the exceptional cases): 7M + 4S. (x2, y2) is a finite affine point in the working form
// (limbs < 2p); *zr, when asked for, receives Z3 / Z1 = 2H (meaningful on the ordinary path only).
__device__ __forceinline__ GJac gj_add_aff29(const GJac &p, const Fr29 &x2, const Fr29 &y2, Fr29 *zr = nullptr) {
    if (gj_is_inf(p)) return GJac{x2, y2, g29_one()};
    const Fr29 Z1Z1 = fr29_sqr(p.Z);                                                      // < 1.03
    const Fr29 U2 = fr29_mul(x2, Z1Z1), S2 = fr29_mul(fr29_mul(y2, p.Z), Z1Z1);                // < 1.02
    const Fr29 H = fr29_norm(fr29_subl(U2, p.X, 1));                                           // < 3.02
    const Fr29 rr = fr29_norm(fr29_subl(S2, p.Y, 1));                                          // < 3.02
    const Fr29 r = fr29_norm(fr29_dbll(rr));                                                   // < 6.04
    const Fr29 HH = fr29_sqr(H);                                                            // < 1.06
    const Fr29 I = fr29_norm(fr29_dbll(fr29_dbll(HH)));                                        // < 4.22
    const Fr29 J = fr29_mul(H, I), V = fr29_mul(p.X, I);                                       // < 1.08, < 1.05
    GJac o;
    o.X = g29_red(fr29_subl(fr29_subl(fr29_sqr(r), J, 1), fr29_norm(fr29_dbll(V)), 2));     // 1.22 + 2 + 4 = 7.22 -> < 2
    // Y3 = r (V - X3) - 2 Y1 J as ONE dot product r (V - X3) + (4p - 2 Y1) J: one Montgomery reduction for the two products, and the result is class A
    // as it leaves the reduction (round 4: 177 instructions of an addition's ~2 700 less than two products, a doubling, a difference and its reduction)
    const Fr29 n2y = fr29_norm(fr29_subl(g29_zero(), fr29_norm(fr29_dbll(p.Y)), 2));           // 4p - 2 Y1 in (0, 4p]
    const Fr29 yl[2] = {r, n2y}, ym[2] = {fr29_norm(fr29_subl(V, o.X, 1)), J};                 // (< 6.04, <= 4) x (< 3.05, < 1.08)
    o.Y = fr29_dot<2>(yl, ym);                                                                 // < 1 + (6.04 * 3.05 + 4 * 1.08) / 169 = 1.14
    o.Z = fr29_mul(fr29_norm(fr29_dbll(p.Z)), H);                                              // 2 Z1 H as a product: 1 + 4 * 3.02 / 169 -> < 1.08 (the squaring form (Z1 + H)^2 - Z1Z1 - HH
                                                                                               // pays a sum, two differences and a reduction for its cheaper multiply)
    if (fr29_is_zero_mod_p(o.Z)) {  // Z1 != 0, so H == 0: same x
        if (fr29_is_zero_mod_p(fr29_lt2p(rr))) return gj_dbl(p);
        return gj_inf();
    }
    if (zr) *zr = g29_red(fr29_dbll(H));                                                       // < 2
    return o;
}
__device__ __forceinline__ GJac gj_add_aff(const GJac &p, const GAff &q) {
    return gj_add_aff29(p, fr29_from(q.x), fr29_from(q.y));                                    // < 1
}
// complete Jacobian addition (add-2007-bl): 11M + 5S
__device__ __forceinline__ GJac gj_add(const GJac &p, const GJac &q) {
    if (gj_is_inf(p)) return q;
    if (gj_is_inf(q)) return p;
    const Fr29 Z1Z1 = fr29_sqr(p.Z), Z2Z2 = fr29_sqr(q.Z);                           // < 1.03
    const Fr29 U1 = fr29_mul(p.X, Z2Z2), U2 = fr29_mul(q.X, Z1Z1);                             // < 1.02
    const Fr29 S1 = fr29_mul(fr29_mul(p.Y, q.Z), Z2Z2), S2 = fr29_mul(fr29_mul(q.Y, p.Z), Z1Z1);
    const Fr29 H = fr29_norm(fr29_subl(U2, U1, 1));                                            // < 3.02
    const Fr29 rr = fr29_norm(fr29_subl(S2, S1, 1));                                           // < 3.02
    const Fr29 r = fr29_norm(fr29_dbll(rr));                                                   // < 6.04
    const Fr29 H2 = fr29_norm(fr29_dbll(H));                                                   // < 6.04
    const Fr29 I = fr29_sqr(H2);                                                           // < 1.22
    const Fr29 J = fr29_mul(H, I), V = fr29_mul(U1, I);                                        // < 1.03, < 1.01
    GJac o;
    o.X = g29_red(fr29_subl(fr29_subl(fr29_sqr(r), J, 1), fr29_norm(fr29_dbll(V)), 2));     // 1.22 + 2 + 4 -> < 2
    const Fr29 n2s = fr29_norm(fr29_subl(g29_zero(), fr29_norm(fr29_dbll(S1)), 2));            // 4p - 2 S1 (as in gj_add_aff29)
    const Fr29 yl[2] = {r, n2s}, ym[2] = {fr29_norm(fr29_subl(V, o.X, 1)), J};
    o.Y = fr29_dot<2>(yl, ym);                                                                 // < 1.14
    const Fr29 zz = fr29_norm(fr29_addl(p.Z, q.Z));                                            // < 4
    const Fr29 T = fr29_norm(fr29_subl(fr29_subl(fr29_sqr(zz), Z1Z1, 1), Z2Z2, 1));        // 1.1 + 4 = 5.1;  = 2 Z1 Z2
    o.Z = fr29_mul(T, H);                                                                      // < 1.1
    if (fr29_is_zero_mod_p(o.Z)) {  // Z1 Z2 != 0, so H == 0
        if (fr29_is_zero_mod_p(fr29_lt2p(rr))) return gj_dbl(p);
        return gj_inf();
    }
    return o;
}
// affine coordinates in the storage form; infinity -> (0, 0) with *inf set (the encoding the restated backend uses;
// unpinned by the reference)
__device__ __forceinline__ GAff gj_to_aff(const GJac &p, bool *inf) {
    *inf = gj_is_inf(p);
    const Fr29 zi = fr29_from(fr_inv(fr29_pack(fr29_canon(p.Z))));  // inverse(0) == 0 -> (0, 0)
    const Fr29 zi2 = fr29_sqr(zi);
    GAff r;
    r.x = fr29_pack(fr29_canon(fr29_mul(p.X, zi2)));
    r.y = fr29_pack(fr29_canon(fr29_mul(p.Y, fr29_mul(zi2, zi))));
    return r;
}

// limb idx (wave-uniform or per-lane) of an 8-limb integer without dynamic register indexing
__device__ __forceinline__ uint32_t limb_at(const Fr &v, uint32_t idx) {
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < 8; k++)
        if ((uint32_t)k == idx) r = v.v[k];
    return r;
}
// bits [pos, pos + n) of a 256-bit integer, zero beyond bit 255 (n <= 16)
__device__ __forceinline__ uint32_t bits_at(const Fr &v, uint32_t pos, uint32_t n) {
    const uint32_t lo = limb_at(v, pos >> 5), hi = limb_at(v, (pos >> 5) + 1);
    const uint64_t two = (uint64_t)hi << 32 | lo;
    return (uint32_t)(two >> (pos & 31)) & ((1u << n) - 1u);
}

// Grumpkin group order q = BN254 base-field modulus (scalar_mul.rs:42-45), little-endian limbs
__device__ __forceinline__ Fr grumpkin_q() {
    return Fr{{0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u}};
}
__device__ __forceinline__ bool int_geq(const Fr &a, const Fr &b) {
    Fr d;
    return fr_sub256(d, a, b) == 0;
}
__device__ __forceinline__ Fr reduce_mod_q(Fr k) {  // 2^256 / q < 6
    for (int it = 0; it < 5; it++) {
        Fr d;
        if (!fr_sub256(d, k, grumpkin_q())) k = d;
    }
    return k;
}

// k * base for a 256-bit integer k with the window tables of `base_index` (0 = G, 1..3 = D[0], D[3], D[6]): 16 mixed additions
// through the 16-bit windows (268 MB of HBM for the four bases) when they were built, else 32 through the 8-bit windows
__device__ __forceinline__ GJac fixed_base_mul(const GrumpkinTables &T, uint32_t base_index, const Fr &k) {
    GJac acc = gj_inf();
    if (T.win16) {
        const uint4 *t16 = T.win16 + (uint64_t)base_index * GRUMPKIN_WIN16_STRIDE * 4;
        // the entries are 64-byte gathers from a 67 MB table (HBM or Infinity Cache latency): the next one is in flight while the
        // current one is added (the address depends on the scalar only)
        auto digit = [&](uint32_t w) { return (limb_at(k, w >> 1) >> (16u * (w & 1u))) & 0xffffu; };
        uint32_t d = digit(0);
        GAff cur = gaff_load(t16, d ? d - 1u : 0u);
        for (uint32_t w = 0; w < 16; w++) {
            const uint32_t dn = w + 1 < 16 ? digit(w + 1) : 0u;
            const GAff nxt = gaff_load(t16, (w + 1 < 16 ? (w + 1) * 65535u : 0u) + (dn ? dn - 1u : 0u));
            if (d) acc = gj_add_aff(acc, cur);
            cur = nxt;
            d = dn;
        }
        return acc;
    }
    const uint4 *tbl = T.win + (uint64_t)base_index * GRUMPKIN_WIN_STRIDE * 4;
    for (uint32_t w = 0; w < 32; w++) {
        const uint32_t d = (limb_at(k, w >> 2) >> (8u * (w & 3u))) & 0xffu;
        if (d) acc = gj_add_aff(acc, gaff_load(tbl, w * 255u + d - 1u));
    }
    return acc;
}

// ------------------------------------------------------------------------------------------------ FixedBaseScalarMul
// [K_FIXED_BASE, opcode, low, high, out_x, fx, out_y, fy]
template <class P>
__device__ __forceinline__ OpResult grumpkin_fixed_base_values(const Fr &low_m, const Fr &high_m, const GrumpkinTables &T, Fr &x, Fr &y) {
    const Fr lo = fr_to_canonical(low_m), hi = fr_to_canonical(high_m);
    if (lo.v[4] | lo.v[5] | lo.v[6] | lo.v[7]) return op_fail_msg(DE_BLACKBOX_FAILED, 10u, DM_LIMB_LOW);    // scalar_mul.rs:25-29
    if (hi.v[4] | hi.v[5] | hi.v[6] | hi.v[7]) return op_fail_msg(DE_BLACKBOX_FAILED, 10u, DM_LIMB_HIGH);
    const Fr k = Fr{{lo.v[0], lo.v[1], lo.v[2], lo.v[3], hi.v[0], hi.v[1], hi.v[2], hi.v[3]}};
    if (int_geq(k, grumpkin_q())) return op_fail_msg(DE_BLACKBOX_FAILED, 10u, DM_SCALAR);                     // scalar_mul.rs:41-51
    bool inf;
    const GAff a = gj_to_aff(fixed_base_mul(T, 0, k), &inf);
    x = a.x;
    y = a.y;
    return op_ok();
}
template <class P>
__device__ __forceinline__ OpResult op_fixed_base(const P &p, const uint32_t *__restrict__ r, const GrumpkinTables &T) {
    if (P::exact) {
        if (!p.known(r[2])) return op_fail(DE_MISSING_ASSIGNMENT, r[2]);
        if (!p.known(r[3])) return op_fail(DE_MISSING_ASSIGNMENT, r[3]);
    }
    Fr x, y;
    const OpResult e = grumpkin_fixed_base_values<P>(p.load(r[2]), p.load(r[3]), T, x, y);
    if (e.err) return e;
    if (!p.insert(r[4], x, r[5])) return op_fail(DE_UNSATISFIED);
    if (!p.insert(r[6], y, r[7])) return op_fail(DE_UNSATISFIED);
    return op_ok();
}

// ------------------------------------------------------------------------------------------------ Pedersen (plookup)
// hash_single: 9-bit slices of the canonical value alternate between two accumulators; slice s = 2i (resp.
// 2i+1) adds (slice + 1) * D[off + i] to accumulator 0 (resp. 1, i < 14); accumulator 0 then takes the endomorphism
// (x, y) -> (beta * x, y).
__device__ __forceinline__ Fr grumpkin_beta() {  // cube root of unity in Fr, Montgomery form of 0xb3c4d79d41a917585bfc41088d8daaa78b17ea66b99c90dd
    Fr c = Fr{{0xb99c90ddu, 0x8b17ea66u, 0x8d8daaa7u, 0x5bfc4108u, 0x41a91758u, 0xb3c4d79du, 0u, 0u}};
    return fr_from_canonical(c);
}
__device__ __forceinline__ GJac pedersen_hash_single(const GrumpkinTables &T, const Fr &v_canon, uint32_t parity) {
    GJac acc0 = gj_inf(), acc1 = gj_inf();
    const uint32_t off = parity ? 15u : 0u;
    for (uint32_t i = 0; i < 15; i++) {
        const uint32_t a = bits_at(v_canon, 18u * i, 9);
        acc0 = gj_add_aff(acc0, gaff_load(T.ped, (off + i) * GRUMPKIN_PED_ENTRIES + a));
        if (i < 14) {
            const uint32_t b = 18u * i + 9u < 256u ? bits_at(v_canon, 18u * i + 9u, 9) : 0u;
            acc1 = gj_add_aff(acc1, gaff_load(T.ped, (off + i) * GRUMPKIN_PED_ENTRIES + b));
        }
    }
    acc0.X = fr29_mul(acc0.X, fr29_from(grumpkin_beta()));
    return gj_add(acc0, acc1);
}
// x coordinate of IV[hash_index] = (hash_index + 1) * G; IV[0].x = G.x = 1
__device__ __forceinline__ Fr pedersen_iv_x(const GrumpkinTables &T, uint32_t hash_index) {
    if (hash_index == 0) return fr_one();
    Fr k = fr_zero();
    k.v[0] = hash_index + 1u;
    k.v[1] = hash_index == 0xFFFFFFFFu ? 1u : 0u;
    bool inf;
    return gj_to_aff(fixed_base_mul(T, 0, k), &inf).x;
}
// one link of the chain: hash_pair(left, right) as an affine point (Montgomery coordinates)
__device__ __forceinline__ GAff pedersen_hash_pair(const GrumpkinTables &T, const Fr &left, const Fr &right) {
    GJac s = gj_inf();
    for (uint32_t parity = 0; parity < 2; parity++) {
        const Fr v = fr_to_canonical(parity ? right : left);
        s = gj_add(s, pedersen_hash_single(T, v, parity));
    }
    bool inf;
    return gj_to_aff(s, &inf);
}
// pedersen(inputs[0..n), hash_index): length-prefixed chain of hash_pairs; inputs are fetched through `get(i)` (Montgomery)
template <class Get>
__device__ __forceinline__ void grumpkin_pedersen(const GrumpkinTables &T, uint32_t n, uint32_t hash_index, Get get, Fr &x, Fr &y) {
    if (n == 0) { x = fr_zero(); y = fr_zero(); return; }
    Fr r = pedersen_iv_x(T, hash_index);
    for (uint32_t step = 0; step <= n; step++) {
        const GAff a = pedersen_hash_pair(T, r, step == 0 ? fr_from_u32(n) : get(step - 1));
        r = a.x;
        y = a.y;
    }
    x = r;
}
// [K_PEDERSEN, opcode, domain_separator, n_in, out_x, fx, out_y, fy, ws...]
template <class P>
__device__ __forceinline__ OpResult op_pedersen(const P &p, const uint32_t *__restrict__ r, const GrumpkinTables &T) {
    const uint32_t n = r[3];
    const uint32_t *ws = r + 8;
    if (P::exact)
        for (uint32_t i = 0; i < n; i++)
            if (!p.known(ws[i])) return op_fail(DE_MISSING_ASSIGNMENT, ws[i]);
    Fr x, y;
    grumpkin_pedersen(T, n, r[2], [&](uint32_t i) { return p.load(ws[i]); }, x, y);
    if (!p.insert(r[4], x, r[5])) return op_fail(DE_UNSATISFIED);
    if (!p.insert(r[6], y, r[7])) return op_fail(DE_UNSATISFIED);
    return op_ok();
}

// ------------------------------------------------------------------------------------------------ SchnorrVerify
// H_j(v) of the hash-ladder Pedersen compress: v (made odd by +1 with a skew correction) = 16 * hi + lo with
// lo odd in [-15, 15]; H_j = hi * D[3j] + lo * D[3j+1] (- D[3j+2] if v was even)
__device__ __forceinline__ GJac ladder_term(const GrumpkinTables &T, const Fr &v_canon, uint32_t j) {
    Fr V = v_canon;
    const bool even = !(V.v[0] & 1u);
    if (even) {  // V + 1 (v < p: no overflow)
        Fr one = fr_zero();
        one.v[0] = 1u;
        fr_add256(V, V, one);
    }
    const int t = (int)((V.v[0] + 16u) & 31u);
    const int lo = t < 16 ? t : t - 32;  // odd
    Fr adj = fr_zero(), hi;
    adj.v[0] = (uint32_t)(lo < 0 ? -lo : lo);
    if (lo >= 0) fr_sub256(hi, V, adj);
    else fr_add256(hi, V, adj);  // V < p < 2^254: no carry out
#pragma unroll
    for (int i = 0; i < 7; i++) hi.v[i] = hi.v[i] >> 4 | hi.v[i + 1] << 28;
    hi.v[7] >>= 4;
    GJac pnt = fixed_base_mul(T, 1u + j, hi);
    GAff q = gaff_load(T.small, j * 15u + adj.v[0] - 1u);
    if (lo < 0) q.y = fr_neg(q.y);
    pnt = gj_add_aff(pnt, q);
    if (even) {
        GAff sk = gaff_load(T.skew, j);
        sk.y = fr_neg(sk.y);
        pnt = gj_add_aff(pnt, sk);
    }
    return pnt;
}

// verify_signature: pk on curve, s and e (mod q) nonzero, R = e * pk + s * G finite, and
// blake2s(be32(compress(R.x, pk.x, pk.y)) || message) == the e bytes of the signature.
// sig / msg bytes come through `sig_byte(i)` / `msg_byte(i)`; the challenge preimage is staged in `m`.
//
// ---- e * P for a per-lane affine point P.
// Schoolbook product of two little-endian 32-bit-limb integers (once per lane: the scalar split below)
template <int N, int M>
__device__ __forceinline__ void bn_mul(const uint32_t (&a)[N], const uint32_t (&b)[M], uint32_t (&r)[N + M]) {
#pragma unroll
    for (int i = 0; i < N + M; i++) r[i] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < M; j++) {
            const uint64_t t = (uint64_t)a[i] * b[j] + r[i + j] + c;
            r[i + j] = (uint32_t)t;
            c = t >> 32;
        }
        r[i + M] = (uint32_t)c;
    }
}
// GLV split for Grumpkin (j-invariant 0): lambda * (x, y) = (beta x, y) with lambda^2 + lambda + 1 = 0 (mod q),
// lambda = 0x59e26bcea0d48bacd4f263f1acdb5c4f5763473177fffffe and beta = grumpkin_beta(). The lattice of (u, v) with
// u + v lambda = 0 (mod q) has the reduced basis (A, -B), (B', A) with determinant q (A = 0x89d3256894d213e2,
// B = 0x6f4d8248eeb859fc8211bbeb7d4f1129, B' = B + 0x89d3256894d213e2 = 0x6f4d...250b). For k < q:
//   c1 = floor(k g1 / 2^384), g1 = floor(2^384 A / q);   c2 = floor(k g2 / 2^384), g2 = floor(2^384 B / q)
//   k1 = k - c1 A - c2 B',   k2 = c1 B - c2 A            =>  k1 + k2 lambda = k (mod q) for ANY integers c1, c2,
// and with these c1, c2 (each within 1 of the exact quotient) |k1|, |k2| < |A| + |B'| < 2^127. Returns the magnitudes
// (4 limbs each) and the signs. Checked against a big-integer model over random and edge scalars (tests/test_gpu_grumpkin.py
// drives it through SchnorrVerify against the oracle's plain double-and-add).
struct GlvSplit {
    uint32_t k1[4], k2[4];
    bool neg1, neg2;
};
__device__ __forceinline__ GlvSplit glv_split(const Fr &k) {
    const uint32_t g1[7] = {0xc85147d0u, 0x5236df9eu, 0x539a2471u, 0x247280eeu, 0xc7e0b3d2u, 0xd91d232eu, 0x00000002u};
    const uint32_t g2[9] = {0x6972c2b8u, 0xa08c1126u, 0x5eaa26e6u, 0xa5e38cfbu, 0x391eb18du, 0x7a7bd9d4u, 0xa773d2cfu, 0x4ccef014u, 0x00000002u};
    const uint32_t A[2] = {0x94d213e2u, 0x89d32568u};
    const uint32_t B[4] = {0x7d4f1129u, 0x8211bbebu, 0xeeb859fcu, 0x6f4d8248u};
    const uint32_t Bp[4] = {0x1221250bu, 0x0be4e154u, 0xeeb859fdu, 0x6f4d8248u};
    uint32_t kk[8];
#pragma unroll
    for (int i = 0; i < 8; i++) kk[i] = k.v[i];
    uint32_t p1[15], p2[17];
    bn_mul<8, 7>(kk, g1, p1);
    bn_mul<8, 9>(kk, g2, p2);
    const uint32_t c1[2] = {p1[12], p1[13]}, c2[4] = {p2[12], p2[13], p2[14], p2[15]};
    uint32_t c1A[4], c2Bp[8], c1B[6], c2A[6];
    bn_mul<2, 2>(c1, A, c1A);
    bn_mul<4, 4>(c2, Bp, c2Bp);
    bn_mul<2, 4>(c1, B, c1B);
    bn_mul<4, 2>(c2, A, c2A);
    auto widen = [](const uint32_t *x, int n) {
        Fr r = fr_zero();
#pragma unroll
        for (int i = 0; i < 8; i++)
            if (i < n) r.v[i] = x[i];
        return r;
    };
    Fr k1, k2, t;
    fr_sub256(t, k, widen(c1A, 4));
    fr_sub256(k1, t, widen(c2Bp, 8));   // two's complement mod 2^256
    fr_sub256(k2, widen(c1B, 6), widen(c2A, 6));
    GlvSplit r;
    r.neg1 = (k1.v[7] >> 31) != 0;
    r.neg2 = (k2.v[7] >> 31) != 0;
    Fr n1, n2;
    fr_sub256(n1, fr_zero(), k1);
    fr_sub256(n2, fr_zero(), k2);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        r.k1[i] = r.neg1 ? n1.v[i] : k1.v[i];
        r.k2[i] = r.neg2 ? n2.v[i] : k2.v[i];
    }
    return r;
}
static constexpr uint32_t VARBASE_TABLE_ENTRIES = 16;  // d P for d = 1..16 (signed 5-bit windows); scratch map: GRUMPKIN_VARBASE_SCRATCH_WORDS (scratch_layout.hpp)
static_assert(GRUMPKIN_VARBASE_SCRATCH_WORDS == VARBASE_TABLE_ENTRIES * 27u + VARBASE_TABLE_ENTRIES * 16u + 4u, "scratch map of grumpkin_var_base_mul");
// Signed 5-bit windows of a magnitude below 2^127: k = sum d_i 32^i with d_i in [-15, 16], i < 26 (a window above 16 borrows 32 from the
// next one). Digit i sits in word i / 5 at bit 6 (i % 5): bits 0..4 the magnitude, bit 5 the sign. 26 windows instead of the 32 unsigned
// 4-bit ones: 52 additions + 130 doublings instead of 64 + 128 for one more table entry.
struct SignedDigits { uint32_t w[6]; };
__device__ __forceinline__ SignedDigits signed_windows5(const uint32_t (&k)[4]) {
    SignedDigits r;
#pragma unroll
    for (int i = 0; i < 6; i++) r.w[i] = 0;
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < 26; i++) {
        const int pos = 5 * i, limb = pos >> 5, sh = pos & 31;
        uint64_t two = k[limb];
        if (limb + 1 < 4) two |= (uint64_t)k[limb + 1] << 32;
        const uint32_t bits = ((uint32_t)(two >> sh) & 31u) + carry;  // 0..32
        const bool neg = bits > 16u;
        const uint32_t mag = neg ? 32u - bits : bits;                // 0..16
        carry = neg ? 1u : 0u;
        r.w[i / 5] |= (mag | (neg ? 32u : 0u)) << (6 * (i % 5));
    }
    return r;  // (k < 2^127: the last window is at most 3 + 1, no carry leaves it)
}
__device__ __forceinline__ uint32_t signed_digit(const SignedDigits &d, uint32_t i) {
    uint32_t word = 0;
#pragma unroll
    for (int q = 0; q < 6; q++)
        if ((uint32_t)q == i / 5u) word = d.w[q];
    return (word >> (6u * (i % 5u))) & 63u;
}
__device__ __forceinline__ uint32_t nibble128(const uint32_t (&k)[4], uint32_t w) {  // 4-bit window w of a 128-bit integer
    uint32_t limb = 0;
#pragma unroll
    for (int i = 0; i < 4; i++)
        if ((uint32_t)i == (w >> 3)) limb = k[i];
    return (limb >> (4u * (w & 7u))) & 15u;
}
// With a window table in the lane's device scratch: e = k1 + k2 lambda (GLV, |k1|, |k2| < 2^127), then 26 joint SIGNED 5-bit windows:
// 5 doublings + the lane's table entry |d1_w| * P + the entry |d2_w| * P mapped through (x, y) -> (beta x, y); a negative half or a negative
// digit negates y. The instruction stream is the same on every lane (bit-serial double-and-add makes the whole wave pay the addition on every
// bit: some lane always has the bit set), and the split halves the doublings: 130 doublings + 52 additions instead of 256 + 64.
// The 16 multiples are brought to ONE denominator so that the 52 additions are mixed ones (11 products instead of 16) without an
// inversion: d P = (X_d, Y_d, Z_d) is built by the chain P, 2P, 2P + P, ... whose every step reports zr_d = Z_d / Z_(d-1); backwards,
// s_d = Z_16 / Z_d = zr_16 ... zr_(d+1) and (X_d s_d^2, Y_d s_d^3, Z_16) is the same point. (x, y) -> (x Z_16^2, y Z_16^3) maps the curve
// onto y^2 = x^3 - 17 Z_16^6, where those pairs are AFFINE points; doubling and addition for a = 0 never read the constant, so the
// whole ladder runs there and the result (X, Y, Z) is the point (X, Y, Z Z_16) of Grumpkin.
// The group has prime order q and 0 < d <= 16, so the chain never meets an exceptional case; the ladder's additions keep theirs.
// Scratch map (`tbl`: a region of GRUMPKIN_VARBASE_SCRATCH_WORDS words per lane, `Bp` lanes, this lane = j):
//   * the chain {X_d, Y_d, zr_d}, 27 words per entry in the working form, word-major [word][lane] like every per-lane buffer: every lane
//     reads and writes the same d at the same time, the accesses are coalesced;
//   * the finished rows {x_d, y_d}, 16 words per entry in the storage form, LANE-major [lane][entry][16 words]: in the ladder every lane
//     reads the row of ITS OWN digit, and a row is one aligned 64-byte line (four 16-byte loads). Word-major, the 18 words of a row were 18
//     scattered 4-byte reads per lane, each of which moved a whole line: 3.0 GB of HBM traffic per 2^16 verifications
//     (profiles/r03r_profile_grumpkin.txt). beta x_d is one product at the addition instead of a third stored field.
// Without a table (Brillig's black-box op): double-and-add.
__device__ __forceinline__ GJac grumpkin_var_base_mul(const GAff &P, const Fr &e, uint32_t *tbl, uint64_t Bp, uint64_t j) {
    GJac a = gj_inf();
    if (!tbl) {
        for (int i = 255; i >= 0; i--) {
            a = gj_dbl(a);
            if ((limb_at(e, (uint32_t)i >> 5) >> (i & 31)) & 1u) a = gj_add_aff(a, P);
        }
        return a;
    }
    auto put = [&](uint32_t d, uint32_t field, const Fr29 &v) {
#pragma unroll
        for (int k = 0; k < 9; k++) tbl[(uint64_t)((d - 1u) * 27u + field * 9u + k) * Bp + j] = v.v[k];
    };
    auto get = [&](uint32_t d, uint32_t field) {
        Fr29 v;
#pragma unroll
        for (int k = 0; k < 9; k++) v.v[k] = tbl[(uint64_t)((d - 1u) * 27u + field * 9u + k) * Bp + j];
        return v;
    };
    // the finished rows: behind the chain, rounded up to 16 bytes (the region holds four words per lane of slack for that)
    uint4 *rows = (uint4 *)(((uintptr_t)(tbl + (uint64_t)(VARBASE_TABLE_ENTRIES * 27u) * Bp) + 15u) & ~(uintptr_t)15u) + j * (VARBASE_TABLE_ENTRIES * 4u);
    auto row_of = [&](uint32_t d) { return rows + (d - 1u) * 4u; };
    // forward: row d - 1 = {X_d, Y_d, zr_d}
    const Fr29 px = fr29_from(P.x), py = fr29_from(P.y);
    GJac q = GJac{px, py, g29_one()};
    put(1, 0, q.X); put(1, 1, q.Y); put(1, 2, q.Z);
    q = gj_dbl(q);                          // Z_2 = 2 y Z_1 = zr_2
    put(2, 0, q.X); put(2, 1, q.Y); put(2, 2, q.Z);
    for (uint32_t d = 3; d <= VARBASE_TABLE_ENTRIES; d++) {
        Fr29 zr = g29_one();
        q = gj_add_aff29(q, px, py, &zr);
        put(d, 0, q.X); put(d, 1, q.Y); put(d, 2, zr);
    }
    const Fr29 z15 = q.Z;  // (the common denominator: Z of the last entry)
    // backward: row d - 1 = {X_d s^2, Y_d s^3} (products: limbs < 2^29, value < 1.4p < 2^256, so the storage form holds them as they are)
    Fr29 sc = g29_one();
    for (uint32_t d = VARBASE_TABLE_ENTRIES; d >= 1; d--) {
        const Fr29 zr = get(d, 2);
        const Fr29 s2 = fr29_sqr(sc);
        const Fr x = fr29_pack(fr29_mul(get(d, 0), s2)), y = fr29_pack(fr29_mul(get(d, 1), fr29_mul(s2, sc)));
        uint4 *row = row_of(d);
        row[0] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
        row[1] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
        row[2] = make_uint4(y.v[0], y.v[1], y.v[2], y.v[3]);
        row[3] = make_uint4(y.v[4], y.v[5], y.v[6], y.v[7]);
        sc = fr29_mul(sc, zr);
    }
    const Fr29 beta = fr29_from(grumpkin_beta());
    const GlvSplit sp = glv_split(e);
    // (requesting both rows of a window before its five doublings -- their index depends on the scalar only -- was measured slower twice: as
    // register loads with the word-major table, 2.30 -> 2.37 ms per 65 536 verifications, and as one-word cache touches with the lane-major
    // rows, 1.70 -> 1.72 ms; so was a per-lane swizzle of the row slots against channel hot spots. The rows are loaded where they are added:
    // the ladder is bound by instruction issue, not by these loads.)
    const SignedDigits d1 = signed_windows5(sp.k1), d2 = signed_windows5(sp.k2);
    for (int w = 25; w >= 0; w--) {
#pragma unroll 1
        for (int k = 0; k < 5; k++) a = gj_dbl(a);  // (one copy of the doubling: with the addition below the loop stays inside the instruction cache)
        for (uint32_t half = 0; half < 2; half++) {  // wave-uniform
            const uint32_t sd = half ? signed_digit(d2, (uint32_t)w) : signed_digit(d1, (uint32_t)w);
            const uint32_t d = sd & 31u;
            const bool neg = (half ? sp.neg2 : sp.neg1) != ((sd & 32u) != 0u);
            if (d) {  // per lane: its own table row
                const uint4 *row = row_of(d);
                const uint4 r0 = row[0], r1 = row[1], r2 = row[2], r3 = row[3];
                Fr29 x = fr29_from(Fr{{r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w}});
                if (half) x = fr29_mul(x, beta);                            // lambda * (x, y) = (beta x, y)
                Fr29 y = fr29_from(Fr{{r2.x, r2.y, r2.z, r2.w, r3.x, r3.y, r3.z, r3.w}});
                const Fr29 ny = fr29_norm(fr29_subl(g29_zero(), y, 1));    // 2p - y: in (0, 2p) since y != 0 (mod p) on this curve
#pragma unroll
                for (int k = 0; k < 9; k++) y.v[k] = neg ? ny.v[k] : y.v[k];
                a = gj_add_aff29(a, x, y);
            }
        }
    }
    a.Z = fr29_lt2p(fr29_mul(a.Z, z15));  // back from the scaled curve; infinity stays infinity
    return a;
}

// s, e: the two halves of the signature as 256-bit integers (big-endian bytes 0..31 and 32..63); the message bytes are staged in `m` behind the
// 32 bytes of the compressed point by `fill_msg(put)`, put(i, byte) storing message byte i
template <class FillMsg>
__device__ __forceinline__ bool grumpkin_schnorr_verify_values(const GrumpkinTables &T, const Fr &pkx, const Fr &pky, Fr s, Fr e, uint32_t n_msg, FillMsg fill_msg, MsgBuf &m,
                                                               uint32_t *window_table = nullptr) {
    const Fr e_raw = e;
    s = reduce_mod_q(s);
    e = reduce_mod_q(e);
    // on curve: y^2 == x^3 - 17
    Fr seventeen = fr_from_u32(17u);
    if (!fr_eq(fr_sqr(pky), fr_sub(fr_mul(fr_sqr(pkx), pkx), seventeen))) return false;
    if (fr_is_zero(s) || fr_is_zero(e)) return false;
    const GJac a = grumpkin_var_base_mul(GAff{pkx, pky}, e, window_table, m.Bp, m.j);
    const GJac b = fixed_base_mul(T, 0, s);
    const GJac rr = gj_add(a, b);
    if (gj_is_inf(rr)) return false;
    bool inf;
    const GAff R = gj_to_aff(rr, &inf);
    // compress(R.x, pk.x, pk.y)
    GJac acc = gj_inf();
    for (uint32_t j = 0; j < 3; j++) {
        const Fr v = fr_to_canonical(j == 0 ? R.x : (j == 1 ? pkx : pky));
        acc = gj_add(acc, ladder_term(T, v, j));
    }
    const Fr c = fr_to_canonical(gj_to_aff(acc, &inf).x);
    m.begin();
    for (uint32_t i = 0; i < 32; i++) m.put(limb_at(c, 7u - (i >> 2)) >> (24u - 8u * (i & 3u)));
    fill_msg([&](uint32_t, uint32_t byte) { m.put(byte); });  // (in order)
    m.end();
    const Digest d = blake2s_msg(m, 32u + n_msg);
    // digest byte i == signature byte 32 + i, i.e. the big-endian bytes of e as given
    uint32_t diff = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) diff |= bswap32(d.d[i]) ^ e_raw.v[7 - i];
    return diff == 0;
}

// the same with the signature and message bytes behind accessors (the Brillig VM's black-box op reads them from its memory)
template <class SigByte, class MsgByte>
__device__ __forceinline__ bool grumpkin_schnorr_verify(const GrumpkinTables &T, const Fr &pkx, const Fr &pky, SigByte sig_byte, uint32_t n_msg,
                                                        MsgByte msg_byte, MsgBuf &m, uint32_t *window_table = nullptr) {
    Fr s = fr_zero(), e = fr_zero();
    for (uint32_t i = 0; i < 32; i++) {  // big-endian 32-byte integers
        const uint32_t sb = sig_byte(i) & 0xffu, eb = sig_byte(32u + i) & 0xffu;
        const uint32_t limb = 7u - (i >> 2), sh = 24u - 8u * (i & 3u);
#pragma unroll
        for (int k = 0; k < 8; k++)
            if ((uint32_t)k == limb) { s.v[k] |= sb << sh; e.v[k] |= eb << sh; }
    }
    return grumpkin_schnorr_verify_values(T, pkx, pky, s, e, n_msg, [&](auto put) { for (uint32_t i = 0; i < n_msg; i++) put(i, msg_byte(i)); }, m, window_table);
}

// [K_SCHNORR, opcode, pkx, pky, n_sig, n_msg, out, flag, sig ws..., msg ws...]
template <class P>
__device__ __forceinline__ OpResult op_schnorr(const P &p, const uint32_t *__restrict__ r, const GrumpkinTables &T, uint32_t *scratch) {
    const uint32_t n_sig = r[4], n_msg = r[5];
    const uint32_t *sig = r + 8, *msg = sig + n_sig;
    if (P::exact) {  // get_inputs_vec order: pkx, pky, signature, message
        if (!p.known(r[2])) return op_fail(DE_MISSING_ASSIGNMENT, r[2]);
        if (!p.known(r[3])) return op_fail(DE_MISSING_ASSIGNMENT, r[3]);
        for (uint32_t i = 0; i < n_sig + n_msg; i++)
            if (!p.known(sig[i])) return op_fail(DE_MISSING_ASSIGNMENT, sig[i]);
    }
    if (n_sig < 64u) return op_fail_msg(DE_PANIC, 5u, DM_SCHNORR_SIG_LEN, n_sig);        // lib.rs:50-52 slice panics
    if (128u + n_msg >= 1024u) return op_fail_msg(DE_PANIC, 5u, DM_SCHNORR_MSG_LEN);      // wasm/schnorr.rs:79-82
    const uint32_t msg_words = (32u + n_msg + 3u) / 4u + 1u;
    MsgBuf m{scratch, p.scratch_stride(), p.scratch_lane(), 0u, 0u};
    // to_u8_vec (signature/mod.rs:5-18): the last big-endian byte of each witness, four rows in flight (ops_common.hpp)
    const Fr s = load_be32_bytes(p, sig), e = load_be32_bytes(p, sig + 32);
    const bool ok = grumpkin_schnorr_verify_values(
        T, p.load(r[2]), p.load(r[3]), s, e, n_msg, [&](auto put) { load_bytes(p, msg, n_msg, put); }, m, scratch + (uint64_t)msg_words * p.scratch_stride());
    if (!p.insert(r[6], ok ? fr_one() : fr_zero(), r[7])) return op_fail(DE_UNSATISFIED);
    return op_ok();
}

template <class P>
__device__ __forceinline__ OpResult dispatch_grumpkin(const P &p, const uint32_t *__restrict__ r, const GrumpkinTables &T, uint32_t *scratch) {
    switch (r[0]) {
    case K_FIXED_BASE: return op_fixed_base(p, r, T);
    case K_PEDERSEN: return op_pedersen(p, r, T);
    case K_SCHNORR: return op_schnorr(p, r, T, scratch);
    default: return op_fail_msg(DE_PANIC, 0, DM_NONE);
    }
}

}  // namespace acvm
