// ops_hash.hpp -- device routines of the hash black boxes, one lane per (instance, call):
//   SHA256 / Blake2s / Keccak256 / Keccak256VariableLength   acvm/src/pwg/blackbox/hash.rs:28-103
//   HashToField128Security                                    acvm/src/pwg/blackbox/hash.rs:13-24
//   the hash functions themselves                             blackbox_solver/src/lib.rs:47-65,86-99
//                                                             (sha2 0.10.7 / blake2 0.10.6 / sha3 0.10.8 Keccak256:
//                                                             FIPS 180-4, RFC 7693, Keccak-f[1600] rate 136 pad 0x01..0x80)
// The message is packed once into a per-lane byte buffer in device scratch (word-major [word][instance], so the 64
// lanes of a wave store and load 256 B contiguous per word) and hashed from there with a per-lane length, which
// covers the variable-length Keccak, the Brillig black-box ops and Schnorr's challenge with the same code.
#pragma once
#include "ops_common.hpp"
#include "hash_device.hpp"

namespace acvm {

// per-lane byte buffer: word wi of instance j at base[wi * Bp + j]
struct MsgBuf {
    uint32_t *base;
    uint64_t Bp, j;
    uint32_t acc, pos;
    __device__ __forceinline__ void begin() { acc = 0; pos = 0; }
    __device__ __forceinline__ void put(uint32_t byte) {
        acc |= (byte & 0xffu) << (8u * (pos & 3u));
        pos++;
        if ((pos & 3u) == 0) { base[(uint64_t)((pos >> 2) - 1) * Bp + j] = acc; acc = 0; }
    }
    __device__ __forceinline__ void end() {
        if (pos & 3u) base[(uint64_t)(pos >> 2) * Bp + j] = acc;
    }
    __device__ __forceinline__ uint32_t word(uint32_t wi) const { return base[(uint64_t)wi * Bp + j]; }
    // little-endian word wi of the message zero-extended beyond len
    __device__ __forceinline__ uint32_t word_le(uint32_t wi, uint32_t len) const {
        if (4u * wi >= len) return 0u;
        uint32_t v = word(wi);
        const uint32_t k = len - 4u * wi;  // valid bytes
        if (k < 4u) v &= (1u << (8u * k)) - 1u;
        return v;
    }
};

// the same message held in LDS by a block that serves 64 instances (kernels_hash.hip hash_coop_level_kernel): word wi of lane l at
// words[wi * 64 + l], byte k of it at byte address 4 * (wi * 64 + l) + k -- conflict-free for both the byte writes and the word reads
struct LdsMsg {
    const uint32_t *words;
    uint32_t lane;
    __device__ __forceinline__ uint32_t word(uint32_t wi) const { return words[wi * 64u + lane]; }
    __device__ __forceinline__ uint32_t word_le(uint32_t wi, uint32_t len) const {
        if (4u * wi >= len) return 0u;
        uint32_t v = word(wi);
        const uint32_t k = len - 4u * wi;  // valid bytes
        if (k < 4u) v &= (1u << (8u * k)) - 1u;
        return v;
    }
};

// one copy per kernel for the lane-per-instance callers (several call sites each in the exact kernels and the Brillig VM)
static inline __device__ __noinline__ Digest sha256_msg(const MsgBuf &m, uint32_t len) { return sha256_body(m, len); }
static inline __device__ __noinline__ Digest blake2s_msg(const MsgBuf &m, uint32_t len) { return blake2s_body(m, len); }
static inline __device__ __noinline__ Digest keccak256_msg(const MsgBuf &m, uint32_t len) { return keccak256_body(m, len); }

// digest bytes as a big-endian integer reduced mod p (from_be_bytes_reduce, generic_ark.rs:281-283), Montgomery form
__device__ __forceinline__ Fr digest_to_field(const Digest &d) {
    Fr c;
#pragma unroll
    for (int i = 0; i < 8; i++) c.v[i] = bswap32(d.d[7 - i]);  // limb i (LE) = big-endian bytes [28 - 4i, 32 - 4i)
    return fr_from_canonical(canon_reduce(c));
}

// ------------------------------------------------------------------------------------------------ the opcode
// [K_HASH, opcode, func, n_in, n_out, var_w, (w, num_bits) x n_in, (out, flag) x n_out]; func = BlackBoxFuncCall tag
// (3 SHA256, 4 Blake2s, 7 HashToField128Security, 11 Keccak256, 12 Keccak256VariableLength), | HASH_COOP_FLAG when every input is one
// byte wide, the function is 3, 4 or 11, n_out == 32 and 1 <= n_in <= 1024 (plan.cpp)
static constexpr uint32_t HASH_COOP_FLAG = 0x100u;
static constexpr uint32_t HASH_CHAIN_FLAG = 0x400u;  // level-schedule copy of the record: one more word behind it, the offset of the link to the record that hashes this digest (kernels_hash.hip)
static constexpr uint32_t HASH_RANGE_FLAG = 0x200u;  // level-schedule copy of the record: (RANGE opcode or NONE, bits) x n_in behind the outputs
template <class P>
__device__ __forceinline__ OpResult op_hash(const P &p, const uint32_t *__restrict__ r, uint32_t *scratch) {
    const uint32_t func = r[2] & 0xffu, n_in = r[3], n_out = r[4], var_w = r[5];  // bit 8: HASH_COOP_FLAG, a hint for the level kernel
    const uint32_t *ins = r + 6, *outs = ins + 2 * n_in;
    if (P::exact) {  // blackbox/mod.rs:55-62, get_inputs_vec order: inputs, then var_message_size
        for (uint32_t i = 0; i < n_in; i++)
            if (!p.known(ins[2 * i])) return op_fail(DE_MISSING_ASSIGNMENT, ins[2 * i]);
        if (var_w != K_NONE && !p.known(var_w)) return op_fail(DE_MISSING_ASSIGNMENT, var_w);
    }
    // get_hash_input (hash.rs:51-86): fetch_nearest_bytes = low ceil(num_bits / 8) bytes, least significant first
    MsgBuf m{scratch, p.scratch_stride(), p.scratch_lane(), 0u, 0u};
    m.begin();
    for (uint32_t i = 0; i < n_in;) {
        const uint32_t nb = (ins[2 * i + 1] + 7u) / 8u;
        if (nb > 32u) return op_fail_msg(DE_PANIC, 0, DM_FETCH_BYTES);  // slice end out of range (generic_ark.rs:316)
        // byte arrays, four witnesses at a time: four rows in flight and four independent reductions for the one wave a SIMD
        // holds at these batch sizes; only the low limb of each canonical value is formed (fr29_redc_low)
        if (nb == 1u && i + 4u <= n_in && (ins[2 * i + 3] + 7u) / 8u == 1u && (ins[2 * i + 5] + 7u) / 8u == 1u && (ins[2 * i + 7] + 7u) / 8u == 1u) {
            const Fr a0 = p.load(ins[2 * i]), a1 = p.load(ins[2 * i + 2]), a2 = p.load(ins[2 * i + 4]), a3 = p.load(ins[2 * i + 6]);
            bool b0, b1, b2, b3;
            const uint32_t l0 = fr_low_limb(a0, b0), l1 = fr_low_limb(a1, b1), l2 = fr_low_limb(a2, b2), l3 = fr_low_limb(a3, b3);
            m.put(l0);
            m.put(l1);
            m.put(l2);
            m.put(l3);
            i += 4u;
            continue;
        }
        const Fr cur = p.load(ins[2 * i]);
        i++;
        if (nb <= 3u) {
            const uint32_t low = fr29_redc_low(fr29_from(cur));
            for (uint32_t k = 0; k < nb; k++) m.put(low >> (8u * k));
            continue;
        }
        const Fr c = fr_to_canonical(cur);
#pragma unroll
        for (int k = 0; k < 32; k++)
            if ((uint32_t)k < nb) m.put(c.v[k >> 2] >> (8 * (k & 3)));
    }
    m.end();
    uint32_t len = m.pos;
    if (var_w != K_NONE) {  // hash.rs:68-83: `to_u128() as usize`
        const Fr c = fr_to_canonical(p.load(var_w));
        if (c.v[1] != 0u || c.v[0] > len) return op_fail_msg(DE_BLACKBOX_FAILED, 11u, DM_KECCAK_VAR_LEN, c.v[0], c.v[1]);
        len = c.v[0];
    }
    Digest d;
    if (func == 3u) d = sha256_msg(m, len);
    else if (func == 4u || func == 7u) d = blake2s_msg(m, len);
    else d = keccak256_msg(m, len);
    if (func == 7u) {  // solve_hash_to_field (hash.rs:13-24)
        if (!p.insert(outs[0], digest_to_field(d), outs[1])) return op_fail(DE_UNSATISFIED);
        return op_ok();
    }
    if (n_out != 32u) return op_fail_msg(DE_BLACKBOX_FAILED, func == 12u ? 11u : func, DM_HASH_OUTPUTS, n_out);  // hash.rs:92-97
    for (uint32_t i = 0; i < 32u; i++)
        if (!p.insert(outs[2 * i], fr_from_byte(d.byte(i)), outs[2 * i + 1])) return op_fail(DE_UNSATISFIED);
    return op_ok();
}

}  // namespace acvm
