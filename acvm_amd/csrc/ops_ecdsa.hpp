// ops_ecdsa.hpp -- ECDSA verification over secp256k1 / secp256r1 on the device, one lane per (instance, call):
//   acvm/src/pwg/blackbox/signature/ecdsa.rs:12-91 (last byte of each witness, length checks, 0 / 1 output) and
//   blackbox_solver/src/lib.rs:66-210 (verify_secp256k1/r1_ecdsa_signature; arithmetic = k256 0.11.6 / p256 0.11.1,
//   i.e. standard ECDSA, SEC 1 v2 section 4.1.4) with the call-site rules of lib.rs: r, s in [1, n-1] or panic; the public
//   key is DECOMPRESSED from x and the parity of y (from_affine_coordinates(.., compress = true)), x >= p or off-curve
//   panics; the digest must be 32 bytes and < n or panic; s > n / 2 ("high S") -> false; R = u1 G + u2 Q, the identity and
//   R.x >= n panic; else R.x == r.
// Both curves have full 256-bit moduli, so this file carries its own generic Montgomery arithmetic (8 x 32-bit limbs with
// the extra carry word) instead of fr_device.hpp's BN254 code. A rare opcode: written for clarity, ~9k products per lane.
#pragma once
#include "ops_common.hpp"

namespace acvm {

struct ModCtx {
    uint32_t m[8], one[8], r2[8], ninv;
    uint32_t r3[8], id;  // R^3 mod m (the way back into Montgomery form behind an inversion), index of the modulus
};
// index: 2 * curve + (0 base field p, 1 group order n); curve 0 = secp256k1, 1 = secp256r1
static __constant__ ModCtx ECDSA_MOD[4] = {
    {{0xfffffc2fu, 0xfffffffeu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu},
     {0x000003d1u, 0x00000001u, 0u, 0u, 0u, 0u, 0u, 0u},
     {0x000e90a1u, 0x000007a2u, 0x00000001u, 0u, 0u, 0u, 0u, 0u}, 0xd2253531u,
     {0x3795f671u, 0x002bb1e3u, 0x00000b73u, 0x00000001u, 0u, 0u, 0u, 0u}, 0u},
    {{0xd0364141u, 0xbfd25e8cu, 0xaf48a03bu, 0xbaaedce6u, 0xfffffffeu, 0xffffffffu, 0xffffffffu, 0xffffffffu},
     {0x2fc9bebfu, 0x402da173u, 0x50b75fc4u, 0x45512319u, 0x00000001u, 0u, 0u, 0u},
     {0x67d7d140u, 0x896cf214u, 0x0e7cf878u, 0x741496c2u, 0x5bcd07c6u, 0xe697f5e4u, 0x81c69bc5u, 0x9d671cd5u}, 0x5588b13fu,
     {0xe9ff41edu, 0x7bc0cfe0u, 0x44d4322cu, 0x00176484u, 0xf1d0b2dau, 0xb1b31347u, 0x18ef116du, 0x555d800cu}, 1u},
    {{0xffffffffu, 0xffffffffu, 0xffffffffu, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000001u, 0xffffffffu},
     {0x00000001u, 0x00000000u, 0x00000000u, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xfffffffeu, 0x00000000u},
     {0x00000003u, 0x00000000u, 0xffffffffu, 0xfffffffbu, 0xfffffffeu, 0xffffffffu, 0xfffffffdu, 0x00000004u}, 0x00000001u,
     {0x0000000au, 0xfffffffdu, 0xfffffff7u, 0xffffffedu, 0xfffffffcu, 0x00000005u, 0x00000001u, 0x00000018u}, 2u},
    {{0xfc632551u, 0xf3b9cac2u, 0xa7179e84u, 0xbce6faadu, 0xffffffffu, 0xffffffffu, 0x00000000u, 0xffffffffu},
     {0x039cdaafu, 0x0c46353du, 0x58e8617bu, 0x43190552u, 0x00000000u, 0x00000000u, 0xffffffffu, 0x00000000u},
     {0xbe79eea2u, 0x83244c95u, 0x49bd6fa6u, 0x4699799cu, 0x2b6bec59u, 0x2845b239u, 0xf3d95620u, 0x66e12d94u}, 0xee00bc4fu,
     {0x0b65a624u, 0xac8ebec9u, 0x0c0555c9u, 0x111f28aeu, 0x6ba5e93fu, 0x2543b924u, 0x6407be65u, 0x503a54e7u}, 3u}};
// curve constants as plain integers: b, Gx, Gy (a = 0 for k1, -3 for r1)
static __constant__ uint32_t ECDSA_CURVE[2][3][8] = {
    {{7u, 0u, 0u, 0u, 0u, 0u, 0u, 0u},
     {0x16f81798u, 0x59f2815bu, 0x2dce28d9u, 0x029bfcdbu, 0xce870b07u, 0x55a06295u, 0xf9dcbbacu, 0x79be667eu},
     {0xfb10d4b8u, 0x9c47d08fu, 0xa6855419u, 0xfd17b448u, 0x0e1108a8u, 0x5da4fbfcu, 0x26a3c465u, 0x483ada77u}},
    {{0x27d2604bu, 0x3bce3c3eu, 0xcc53b0f6u, 0x651d06b0u, 0x769886bcu, 0xb3ebbd55u, 0xaa3a93e7u, 0x5ac635d8u},
     {0xd898c296u, 0xf4a13945u, 0x2deb33a0u, 0x77037d81u, 0x63a440f2u, 0xf8bce6e5u, 0xe12c4247u, 0x6b17d1f2u},
     {0x37bf51f5u, 0xcbb64068u, 0x6b315eceu, 0x2bce3357u, 0x7c0f9e16u, 0x8ee7eb4au, 0xfe1a7f9bu, 0x4fe342e2u}}};

// a 256-bit integer / residue travels in the 8-limb Fr container
__device__ __forceinline__ Fr mc_limbs(const uint32_t *p) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = p[i];
    return r;
}
__device__ __forceinline__ bool u_geq(const Fr &a, const Fr &b) {
    Fr d;
    return fr_sub256(d, a, b) == 0;
}
// Montgomery product for a full 256-bit odd modulus: CIOS with the extra carry word
__device__ __forceinline__ Fr mm_mul(const Fr &a, const Fr &b, const ModCtx &f) {
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
        const uint32_t m = t[0] * f.ninv;
        c = ((uint64_t)m * f.m[0] + t[0]) >> 32;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            c += (uint64_t)m * f.m[j] + t[j];
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
    const Fr mod = mc_limbs(f.m);
    const uint32_t borrow = fr_sub256(d, r, mod);
    const bool sub = t[8] != 0u || borrow == 0u;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = sub ? d.v[i] : r.v[i];
    return r;
}
__device__ __forceinline__ Fr mm_add(const Fr &a, const Fr &b, const ModCtx &f) {
    Fr r, d;
    const uint32_t c = fr_add256(r, a, b);
    const uint32_t borrow = fr_sub256(d, r, mc_limbs(f.m));
    const bool sub = c != 0u || borrow == 0u;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = sub ? d.v[i] : r.v[i];
    return r;
}
__device__ __forceinline__ Fr mm_sub(const Fr &a, const Fr &b, const ModCtx &f) {
    Fr r, q;
    const uint32_t mask = fr_sub256(r, a, b) ? 0xffffffffu : 0u;
#pragma unroll
    for (int i = 0; i < 8; i++) q.v[i] = f.m[i] & mask;
    fr_add256(r, r, q);
    return r;
}
__device__ __forceinline__ Fr mm_to(const Fr &a, const ModCtx &f) { return mm_mul(a, mc_limbs(f.r2), f); }
__device__ __forceinline__ Fr mm_from(const Fr &a, const ModCtx &f) {
    Fr o = fr_zero();
    o.v[0] = 1u;
    return mm_mul(a, o, f);
}
// a^e for a wave-uniform exponent (p - 2, n - 2, (p + 1) / 4: constants of the curve): fixed 4-bit windows, 256 squarings + 64 products + 14 for
// the table instead of a product per set bit (p - 2 of secp256k1 has 250 of them). The table is indexed by the window, which every lane shares.
static inline __device__ __noinline__ Fr mm_pow(const Fr &a, const Fr &e, const ModCtx &f) {
    Fr tab[16];
    tab[0] = mc_limbs(f.one);
    tab[1] = a;
    for (int k = 2; k < 16; k++) tab[k] = mm_mul(tab[k - 1], a, f);
    Fr acc = mc_limbs(f.one);
    for (int i = 63; i >= 0; i--) {
        acc = mm_mul(acc, acc, f);
        acc = mm_mul(acc, acc, f);
        acc = mm_mul(acc, acc, f);
        acc = mm_mul(acc, acc, f);
        uint32_t w = 0;
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (k == (i >> 3)) w = e.v[k];
        const uint32_t d = __builtin_amdgcn_readfirstlane((w >> (4 * (i & 7))) & 15u);
        if (d) acc = mm_mul(acc, tab[d], f);
    }
    return acc;
}
// Inversion by safegcd (fr_device.hpp fr_safegcd_inv, the divstep schedule of libsecp256k1's modinv32) instead of the exponentiation a^(m - 2):
// ~19 k issue slots against 334 Montgomery products (~100 k). The routine is instantiated per modulus (its 9 x 30-bit limbs and m^-1 mod 2^30
// are compile-time constants); the modulus of a call is wave-uniform.
template <int K>
struct EcMod30 {
    static __device__ __forceinline__ int32_t p30(int i) {
        constexpr int32_t L[4][9] = {
            {0x3ffffc2f, 0x3ffffffb, 0x3fffffff, 0x3fffffff, 0x3fffffff, 0x3fffffff, 0x3fffffff, 0x3fffffff, 0xffff},
            {0x10364141, 0x3f497a33, 0x348a03bb, 0x2bb739ab, 0x3ffffeba, 0x3fffffff, 0x3fffffff, 0x3fffffff, 0xffff},
            {0x3fffffff, 0x3fffffff, 0x3fffffff, 0x0000003f, 0x00000000, 0x00000000, 0x00001000, 0x3fffc000, 0xffff},
            {0x3c632551, 0x0ee72b0b, 0x3179e84f, 0x39beab69, 0x3fffffbc, 0x3fffffff, 0x00000fff, 0x3fffc000, 0xffff}};
        return L[K][i];
    }
    static constexpr uint32_t PINV30 = K == 0 ? 0x2ddacacfu : K == 1 ? 0x2a774ec1u : K == 2 ? 0x3fffffffu : 0x11ff43b1u;
};
template <int K>
static inline __device__ __noinline__ Fr ec_safegcd(const Fr &a) { return fr_safegcd_inv<EcMod30<K>>(a); }
// a^-1 in Montgomery form for a in Montgomery form (0 for 0): the integer a R inverts to a^-1 R^-1, one product with R^3 returns a^-1 R
__device__ __forceinline__ Fr mm_inv(const Fr &a, const ModCtx &f) {
    Fr x;
    switch (f.id) {
    case 0: x = ec_safegcd<0>(a); break;
    case 1: x = ec_safegcd<1>(a); break;
    case 2: x = ec_safegcd<2>(a); break;
    default: x = ec_safegcd<3>(a); break;
    }
    return mm_mul(x, mc_limbs(f.r3), f);
}

struct EJac { Fr X, Y, Z; };  // Montgomery coordinates mod p, Z == 0 <=> identity
__device__ __forceinline__ EJac ej_identity(const ModCtx &f) { return EJac{mc_limbs(f.one), mc_limbs(f.one), fr_zero()}; }
// a = 0 (secp256k1) or a = -3 (secp256r1): M = 3 X^2 + a Z^4
__device__ __forceinline__ EJac ej_dbl(const EJac &p, const ModCtx &f, bool a_minus3) {
    if (fr_is_zero(p.Z) || fr_is_zero(p.Y)) return ej_identity(f);
    const Fr yy = mm_mul(p.Y, p.Y, f), yyyy = mm_mul(yy, yy, f);
    Fr s = mm_mul(p.X, yy, f);
    s = mm_add(s, s, f);
    s = mm_add(s, s, f);
    Fr m;
    if (a_minus3) {  // 3 X^2 - 3 Z^4 = 3 (X - Z^2)(X + Z^2): one squaring and one product instead of three squarings
        const Fr zz = mm_mul(p.Z, p.Z, f);
        m = mm_mul(mm_sub(p.X, zz, f), mm_add(p.X, zz, f), f);
    } else m = mm_mul(p.X, p.X, f);
    m = mm_add(mm_add(m, m, f), m, f);
    EJac r;
    r.X = mm_sub(mm_sub(mm_mul(m, m, f), s, f), s, f);
    Fr y8 = mm_add(yyyy, yyyy, f);
    y8 = mm_add(y8, y8, f);
    y8 = mm_add(y8, y8, f);
    r.Y = mm_sub(mm_mul(m, mm_sub(s, r.X, f), f), y8, f);
    const Fr yz = mm_mul(p.Y, p.Z, f);
    r.Z = mm_add(yz, yz, f);
    return r;
}
// complete mixed addition: q = (x, y) a finite affine point (Montgomery coordinates); 8M + 3S
struct EAff { Fr x, y; };
__device__ __forceinline__ EJac ej_add_aff(const EJac &p, const EAff &q, const ModCtx &f, bool a_minus3) {
    if (fr_is_zero(p.Z)) return EJac{q.x, q.y, mc_limbs(f.one)};
    const Fr z1z1 = mm_mul(p.Z, p.Z, f);
    const Fr u2 = mm_mul(q.x, z1z1, f), s2 = mm_mul(mm_mul(q.y, p.Z, f), z1z1, f);
    const Fr h = mm_sub(u2, p.X, f), rr = mm_sub(s2, p.Y, f);
    if (fr_is_zero(h)) return fr_is_zero(rr) ? ej_dbl(p, f, a_minus3) : ej_identity(f);
    const Fr hh = mm_mul(h, h, f), hhh = mm_mul(hh, h, f), v = mm_mul(p.X, hh, f);
    EJac r;
    r.X = mm_sub(mm_sub(mm_sub(mm_mul(rr, rr, f), hhh, f), v, f), v, f);
    r.Y = mm_sub(mm_mul(rr, mm_sub(v, r.X, f), f), mm_mul(p.Y, hhh, f), f);
    r.Z = mm_mul(p.Z, h, f);
    return r;
}
// u1 G + u2 Q on ONE ladder (Shamir): 256 doublings, and per bit pair the addition of G, Q or G + Q -- all three affine (one inversion for
// G + Q), so every lane runs the same mixed addition whatever its bits are. Two separate double-and-add ladders with full Jacobian additions
// cost 512 doublings + 512 additions wave-wide (some lane always has the bit set): 16.5 ms per 65 536 verifications; this one 7.x ms.
static inline __device__ __noinline__ EJac ej_mul2(const EAff &G, const Fr &u1, const EAff &Q, const Fr &u2, const ModCtx &f, bool a_minus3) {
    const EJac gq = ej_add_aff(EJac{G.x, G.y, mc_limbs(f.one)}, Q, f, a_minus3);
    const bool gq_inf = fr_is_zero(gq.Z);  // Q == -G: the pair (1, 1) adds nothing
    const Fr zi = mm_inv(gq.Z, f), zi2 = mm_mul(zi, zi, f);
    const EAff GQ{mm_mul(gq.X, zi2, f), mm_mul(gq.Y, mm_mul(zi2, zi, f), f)};
    EJac acc = ej_identity(f);
    for (int i = 255; i >= 0; i--) {
        acc = ej_dbl(acc, f, a_minus3);
        uint32_t w1 = 0, w2 = 0;
#pragma unroll
        for (int kk = 0; kk < 8; kk++)
            if (kk == (i >> 5)) { w1 = u1.v[kk]; w2 = u2.v[kk]; }
        const uint32_t sel = ((w1 >> (i & 31)) & 1u) | (((w2 >> (i & 31)) & 1u) << 1);
        EAff e;
#pragma unroll
        for (int kk = 0; kk < 8; kk++) {
            e.x.v[kk] = sel == 1u ? G.x.v[kk] : (sel == 2u ? Q.x.v[kk] : GQ.x.v[kk]);
            e.y.v[kk] = sel == 1u ? G.y.v[kk] : (sel == 2u ? Q.y.v[kk] : GQ.y.v[kk]);
        }
        if (sel != 0u && !(sel == 3u && gq_inf)) acc = ej_add_aff(acc, e, f, a_minus3);
    }
    return acc;
}

// panic codes (host texts in batch.cpp ecdsa_panic_text)
enum EcdsaPanic : uint32_t { EP_SIG = 1, EP_PUBKEY = 2, EP_MSG_LEN = 3, EP_MSG_RANGE = 4, EP_IDENTITY = 5, EP_X_RANGE = 6 };

// 1 valid, 0 invalid, or a panic code in *panic. Bytes come through byte accessors (32 + 32 + 64 + n_msg).
template <class PkX, class PkY, class Sig, class Msg>
__device__ __forceinline__ uint32_t ecdsa_verify(uint32_t curve, PkX pkx, PkY pky, Sig sig, uint32_t n_msg, Msg msg, uint32_t *panic) {
    const ModCtx &fp = ECDSA_MOD[2 * curve], &fn = ECDSA_MOD[2 * curve + 1];
    const bool a_minus3 = curve == 1u;
    auto be32 = [&](auto get, uint32_t off) {  // 32 big-endian bytes -> limbs
        Fr r = fr_zero();
        for (uint32_t i = 0; i < 32; i++) {
            const uint32_t byte = get(off + i) & 0xffu, limb = 7u - (i >> 2), sh = 24u - 8u * (i & 3u);
#pragma unroll
            for (int k = 0; k < 8; k++)
                if ((uint32_t)k == limb) r.v[k] |= byte << sh;
        }
        return r;
    };
    *panic = 0;
    const Fr n = mc_limbs(fn.m), p = mc_limbs(fp.m);
    const Fr r = be32(sig, 0), s = be32(sig, 32);
    if (fr_is_zero(r) || fr_is_zero(s) || u_geq(r, n) || u_geq(s, n)) { *panic = EP_SIG; return 0; }
    const Fr x = be32(pkx, 0);
    if (u_geq(x, p)) { *panic = EP_PUBKEY; return 0; }
    const Fr xm = mm_to(x, fp);
    Fr rhs = mm_mul(mm_mul(xm, xm, fp), xm, fp);
    if (a_minus3) rhs = mm_sub(rhs, mm_add(mm_add(xm, xm, fp), xm, fp), fp);
    rhs = mm_add(rhs, mm_to(mc_limbs(ECDSA_CURVE[curve][0]), fp), fp);
    // p = 3 mod 4 on both curves: sqrt = rhs^((p + 1) / 4)
    Fr e = p, one_i = fr_zero();
    one_i.v[0] = 1u;
    fr_add256(e, e, one_i);
#pragma unroll
    for (int i = 0; i < 7; i++) e.v[i] = e.v[i] >> 2 | e.v[i + 1] << 30;
    e.v[7] >>= 2;
    Fr ym = mm_pow(rhs, e, fp);
    if (!fr_eq(mm_mul(ym, ym, fp), rhs)) { *panic = EP_PUBKEY; return 0; }
    if ((mm_from(ym, fp).v[0] & 1u) != (pky(31) & 1u)) ym = mm_sub(fr_zero(), ym, fp);
    if (n_msg != 32u) { *panic = EP_MSG_LEN; return 0; }
    const Fr z = be32(msg, 0);
    if (u_geq(z, n)) { *panic = EP_MSG_RANGE; return 0; }
    Fr half;
#pragma unroll
    for (int i = 0; i < 7; i++) half.v[i] = n.v[i] >> 1 | n.v[i + 1] << 31;
    half.v[7] = n.v[7] >> 1;
    Fr d;
    if (fr_sub256(d, half, s)) return 0;  // s > n / 2: not low-S normalised
    const Fr si = mm_inv(mm_to(s, fn), fn);
    const Fr u1 = mm_from(mm_mul(mm_to(z, fn), si, fn), fn), u2 = mm_from(mm_mul(mm_to(r, fn), si, fn), fn);
    const EJac R = ej_mul2(EAff{mm_to(mc_limbs(ECDSA_CURVE[curve][1]), fp), mm_to(mc_limbs(ECDSA_CURVE[curve][2]), fp)}, u1, EAff{xm, ym}, u2, fp, a_minus3);
    if (fr_is_zero(R.Z)) { *panic = EP_IDENTITY; return 0; }
    const Fr zi = mm_inv(R.Z, fp);
    const Fr rx = mm_from(mm_mul(R.X, mm_mul(zi, zi, fp), fp), fp);
    if (u_geq(rx, n)) { *panic = EP_X_RANGE; return 0; }
    return fr_eq(rx, r) ? 1u : 0u;
}

// [K_ECDSA, opcode, curve, n_x, n_y, n_sig, n_msg, out, flag, ws: x..., y..., sig..., msg...]
template <class P>
__device__ __forceinline__ OpResult op_ecdsa(const P &p, const uint32_t *__restrict__ r) {
    const uint32_t curve = r[2], n_x = r[3], n_y = r[4], n_sig = r[5], n_msg = r[6];
    const uint32_t *wx = r + 9, *wy = wx + n_x, *wsig = wy + n_y, *wmsg = wsig + n_sig;
    const uint32_t func = 8u + curve;  // BlackBoxFunc::EcdsaSecp256k1 / EcdsaSecp256r1
    if (P::exact)  // get_inputs_vec order: public_key_x, public_key_y, signature, hashed_message
        for (uint32_t i = 0; i < n_x + n_y + n_sig + n_msg; i++)
            if (!p.known(wx[i])) return op_fail(DE_MISSING_ASSIGNMENT, wx[i]);
    // ecdsa.rs:20-47: length checks in the order pubkey_x, pubkey_y, signature
    if (n_x != 32u) return op_fail_msg(DE_BLACKBOX_FAILED, func, DM_ECDSA_LEN, 0u, n_x);
    if (n_y != 32u) return op_fail_msg(DE_BLACKBOX_FAILED, func, DM_ECDSA_LEN, 1u, n_y);
    if (n_sig != 64u) return op_fail_msg(DE_BLACKBOX_FAILED, func, DM_ECDSA_LEN, 2u, n_sig);
    auto byte_of = [&](const uint32_t *ws, uint32_t i) { return fr_to_canonical(p.load(ws[i])).v[0] & 0xffu; };  // to_u8_vec
    uint32_t panic = 0;
    const uint32_t ok = ecdsa_verify(
        curve, [&](uint32_t i) { return byte_of(wx, i); }, [&](uint32_t i) { return byte_of(wy, i); }, [&](uint32_t i) { return byte_of(wsig, i); }, n_msg,
        [&](uint32_t i) { return byte_of(wmsg, i); }, &panic);
    if (panic) return OpResult{DE_PANIC, func, panic, DM_ECDSA_PANIC, panic, 0u};
    if (!p.insert(r[7], ok ? fr_one() : fr_zero(), r[8])) return op_fail(DE_UNSATISFIED);
    return op_ok();
}

}  // namespace acvm
