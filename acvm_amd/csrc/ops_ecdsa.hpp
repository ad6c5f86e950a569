// ops_ecdsa.hpp -- ECDSA verification over secp256k1 / secp256r1 on the device, one lane per (instance, call):
//   acvm/src/pwg/blackbox/signature/ecdsa.rs:12-91 (last byte of each witness, length checks, 0 / 1 output) and
//   blackbox_solver/src/lib.rs:66-210 (verify_secp256k1/r1_ecdsa_signature; arithmetic = k256 0.11.6 / p256 0.11.1,
//   i.e. standard ECDSA, SEC 1 v2 section 4.1.4) with the call-site rules of lib.rs: r, s in [1, n-1] or panic; the public
//   key is DECOMPRESSED from x and the parity of y (from_affine_coordinates(.., compress = true)), x >= p or off-curve
//   panics; the digest must be 32 bytes and < n or panic; s > n / 2 ("high S") -> false; R = u1 G + u2 Q, the identity and
//   R.x >= n panic; else R.x == r.
// The curve arithmetic (prime-shaped reductions, window tables, the order of these checks) is secp_device.hpp; this file reads the bytes.
#pragma once
#include "ops_common.hpp"
#include "secp_device.hpp"

namespace acvm {

// 1 valid, 0 invalid, or a panic code in *panic. Bytes come through byte accessors (32 + 32 + 64 + n_msg). gtab: the generator tables of both
// curves (kernels_ecdsa.hip ecdsa_generator_tables), curve c at c * SECP_GTABLE_WORDS.
// the same on assembled integers: r, s (signature halves), x (public key), the parity of y, the digest z (n_msg == 32, else it panics inside)
__device__ __forceinline__ uint32_t ecdsa_verify_values(uint32_t curve, const Fr &r, const Fr &s, const Fr &x, const Fr &y, uint32_t n_msg, const Fr &z,
                                                        const uint32_t *__restrict__ gtab, uint32_t *panic) {
    const uint32_t y_odd = y.v[0] & 1u;  // (the last byte of public_key_y)
    return curve == 0u ? secp_verify<0>(r, s, x, y_odd, n_msg, z, gtab, panic, &y) : secp_verify<1>(r, s, x, y_odd, n_msg, z, gtab + SECP_GTABLE_WORDS, panic, &y);
}
template <class PkX, class PkY, class Sig, class Msg>
__device__ __forceinline__ uint32_t ecdsa_verify(uint32_t curve, PkX pkx, PkY pky, Sig sig, uint32_t n_msg, Msg msg, const uint32_t *__restrict__ gtab,
                                                 uint32_t *panic) {
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
    const Fr r = be32(sig, 0), s = be32(sig, 32), x = be32(pkx, 0), y = be32(pky, 0);
    const Fr z = n_msg == 32u ? be32(msg, 0) : fr_zero();
    return ecdsa_verify_values(curve, r, s, x, y, n_msg, z, gtab, panic);
}

// [K_ECDSA, opcode, curve, n_x, n_y, n_sig, n_msg, out, flag, ws: x..., y..., sig..., msg...]
template <class P>
__device__ __forceinline__ OpResult op_ecdsa(const P &p, const uint32_t *__restrict__ r, const uint32_t *__restrict__ gtab) {
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
    // to_u8_vec, 32 witnesses to a 256-bit integer, four rows in flight (ops_common.hpp load_be32_bytes); the reference uses the parity of y only,
    // the whole of it is read here to spare the square root of the decompression (secp_device.hpp secp_verify)
    uint32_t panic = 0;
    const Fr sr = load_be32_bytes(p, wsig), ss = load_be32_bytes(p, wsig + 32), x = load_be32_bytes(p, wx), y = load_be32_bytes(p, wy);
    const Fr z = n_msg == 32u ? load_be32_bytes(p, wmsg) : fr_zero();
    const uint32_t ok = ecdsa_verify_values(curve, sr, ss, x, y, n_msg, z, gtab, &panic);
    if (panic) return OpResult{DE_PANIC, func, panic, DM_ECDSA_PANIC, panic, 0u};
    if (!p.insert(r[7], ok ? fr_one() : fr_zero(), r[8])) return op_fail(DE_UNSATISFIED);
    return op_ok();
}

}  // namespace acvm
