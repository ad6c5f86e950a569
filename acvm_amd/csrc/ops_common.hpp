// ops_common.hpp -- device-side scaffolding shared by every non-arithmetic opcode kernel and by the exact
// in-order kernel: record kinds, the two execution policies, expression evaluation (pwg/mod.rs:321-372 get_value,
// arithmetic.rs:212-239 evaluate) and canonical-integer helpers.
//
// Every opcode's device routine is written once, templated on a Policy:
//   FastPolicy  -- the generic instance of plan.cpp: every witness the planner saw as assigned is assigned; an output
//                  the planner saw as already assigned is compared, never overwritten (insert_value conflict,
//                  pwg/mod.rs:338-357); any error only flags the instance (event word) for the exact kernel.
//   ExactPolicy -- per-instance assigned bitmap, exact error kind / aux values.
#pragma once
#include "fr_device.hpp"
#include "gate_eval.hpp"

namespace acvm {

static constexpr uint32_t K_COEF_ONE = 0xFFFFFFFFu;
static constexpr uint32_t K_COEF_MINUS_ONE = 0xFFFFFFFEu;
static constexpr uint32_t K_COEF_ZERO = 0xFFFFFFFDu;
static constexpr uint32_t K_NONE = 0xFFFFFFFFu;

// record kinds of the in-order program (w0 of each record; w1 = unused for ARITH, see plan.cpp)
enum RecKind : uint32_t {
    K_ARITH = 0, K_RANGE = 1, K_LOGIC = 2, K_HASH = 3, K_PEDERSEN = 4, K_FIXED_BASE = 5, K_SCHNORR = 6, K_ZERO_OUT = 7,
    K_QUOTIENT = 8, K_TO_LE_RADIX = 9, K_MEM_INIT = 10, K_MEM_OP = 11, K_BRILLIG = 12, K_ECDSA = 13, K_PERM_SORT = 14,
    K_DIGEST_LEAF = 15, K_RANGE_MULTI = 16, K_BRILLIG_SL = 17  // level-schedule records without an opcode of their own (plan.cpp)
};

// error codes = ACVM_ERR_* of include/acvm_amd.h
enum DevErr : uint32_t {
    DE_NONE = 0, DE_MISSING_ASSIGNMENT = 1, DE_TOO_MANY_UNKNOWNS = 2, DE_UNSUPPORTED_BLACKBOX = 3, DE_UNSATISFIED = 4,
    DE_INDEX_OOB = 5, DE_BLACKBOX_FAILED = 6, DE_BRILLIG_FAILED = 7, DE_PANIC = 8,
    DE_WAIT_FOREIGN_CALL = 100  // not an error: Brillig needs a foreign call result (x0 = bytecode index of the ForeignCall)
};
// sub-codes carried in aux1 for errors whose message text is rebuilt on the host
enum DevMsg : uint32_t {
    DM_NONE = 0, DM_TWO_MUL_TERMS = 1, DM_LOGIC_BITS = 2, DM_FETCH_BYTES = 3, DM_HASH_OUTPUTS = 4, DM_KECCAK_VAR_LEN = 5,
    DM_MEM_INDEX_U64 = 6, DM_MEM_READ_EXPR = 7, DM_RADIX = 8, DM_LIMB_LOW = 9, DM_LIMB_HIGH = 10, DM_SCALAR = 11,
    DM_SCHNORR_SIG_LEN = 12, DM_SCHNORR_MSG_LEN = 13, DM_BRILLIG_TRAP = 14, DM_BRILLIG_RETURN = 15, DM_BRILLIG_PANIC = 16,
    DM_BRILLIG_MEM_CAP = 17, DM_BRILLIG_STEP_LIMIT = 18, DM_BRILLIG_BB_FAILED = 19, DM_PEDERSEN_DOMAIN = 20, DM_FC_COUNT = 21,
    DM_FC_SIZE = 22, DM_FC_PENDING_CAP = 23, DM_HOST_MESSAGE = 24, DM_ECDSA_LEN = 25, DM_ECDSA_PANIC = 26, DM_SORT_TUPLE = 27, DM_BRILLIG_CALL_DEPTH = 28,
    DM_DEVICE_LIMIT = 29  // host only: an instance the exact path gave up on (batch.cpp retry_device_limits)
};

// err / aux0 / aux1 are the ABI's acvm_result_t fields; msg (DevMsg) and x0, x1 let the host rebuild the message text
struct OpResult {
    uint32_t err, aux0, aux1, msg, x0, x1;
};
__device__ __forceinline__ OpResult op_ok() { return OpResult{DE_NONE, 0u, 0u, 0u, 0u, 0u}; }
__device__ __forceinline__ OpResult op_fail(uint32_t e, uint32_t a0 = 0, uint32_t a1 = 0) { return OpResult{e, a0, a1, 0u, 0u, 0u}; }
__device__ __forceinline__ OpResult op_fail_msg(uint32_t e, uint32_t a0, uint32_t msg, uint32_t x0 = 0, uint32_t x1 = 0) {
    return OpResult{e, a0, 0u, msg, x0, x1};
}

__device__ __forceinline__ Fr apply_coef(const Fr &x, uint32_t coef, const uint32_t *__restrict__ consts) {
    if (coef == K_COEF_ONE) return x;
    if (coef == K_COEF_MINUS_ONE) return fr_neg(x);
    return fr_mul(x, fr_const(consts, coef));
}
// coef * (a * b) and coef * x without leaving the 29-bit working form between the two products: the intermediate is
// neither reduced nor repacked (the product tolerates inputs < 8p)
__device__ __forceinline__ Fr apply_coef_prod(const Fr &a, const Fr &b, uint32_t coef, const uint32_t *__restrict__ consts) {
    const Fr29 t = fr29_mul(fr29_from(a), fr29_from(b));
    if (coef == K_COEF_ONE) return fr29_pack(fr29_cond_sub_p(t));
    if (coef == K_COEF_MINUS_ONE) return fr_neg(fr29_pack(fr29_cond_sub_p(t)));
    return fr29_pack(fr29_cond_sub_p(fr29_mul(t, fr29_from(fr_const(consts, coef)))));
}
// ---- the event words of a batch: event[j] = the first opcode (program order) at which instance j left the generic path, 0xFFFFFFFF = it did
// not. In FRONT of them, four words: [-4] how many instances are flagged, [-3] spare, [-2..-1] the address of a host-mapped counter (or null).
// The FIRST flag of an instance counts it, on the device (the gate of the next tile's import reads that word) and in the host's counter (the
// host reads it after the synchronisation it needs anyway): no counting kernel behind a solve (until round 6: event_count_kernel).
static constexpr int EVENT_HDR_WORDS = 4;
// (The lanes of a wave that flag together count together: one pair of atomics per wave, not per instance -- a batch in which every instance fails at
// the same opcode would otherwise send 2^17 system-scope atomics over PCIe to one host word.)
__device__ __forceinline__ void flag_instance(uint32_t *__restrict__ event, uint64_t j, uint32_t opcode) {
    const bool first = atomicMin(&event[j], opcode) == 0xFFFFFFFFu;
    const uint64_t firsts = __builtin_amdgcn_ballot_w64(first);  // over the lanes that are here (the callers sit inside `if (error)`)
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    if (first && (uint32_t)__builtin_ctzll(firsts) == lane) {
        const uint32_t n = (uint32_t)__builtin_popcountll(firsts);
        atomicAdd(event - 4, n);
        uint32_t *host = *(uint32_t *const *)(event - 2);
        if (host) __hip_atomic_fetch_add(host, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// ---- the gate kernel's body (kernels.hip arith_level_kernel; kernels_ops.hip runs it beside the light records of the same level):
// gate_eval.hpp's record evaluation over the witness table. grid = (ceil(B/256), gates in level). Lane = instance. The gate record
// (number gate_index of the level) is wave-uniform.
struct GateDeviceLoader {
    const uint4 *__restrict__ W;
    const uint4 *__restrict__ Inv;
    const uint32_t *__restrict__ consts;
    uint64_t Bp, j;
    __device__ __forceinline__ Fr29 load(uint32_t slot) const {
        // nontemporal (streaming) loads: an operand row is read by this launch and then not again for levels (5.53 -> 5.55 M witnesses/s)
#ifdef GATE_EXP_NO_LOADS  // measurement only (tools/build_variant.sh): operands made up from the lane and the row, no memory access
        Fr29 r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.v[i] = ((uint32_t)j * 2654435761u + slot * 40503u + i) & 0x1fffffffu;
        r.v[8] &= 0xffffffu;
        return r;
#else
        return fr29_from(fr_load_nt(W, slot, Bp, j));
#endif
    }
    __device__ __forceinline__ Fr29 load_inverse(uint32_t slot) const { return fr29_from(fr_load_nt(Inv, slot, Bp, j)); }
    __device__ __forceinline__ GateWords constant(uint32_t idx) const { return GATE_WORDS(consts) + (uint64_t)idx * 8; }
    __device__ __forceinline__ bool any(bool x) const { return __builtin_amdgcn_ballot_w64(x) != 0; }
};
__device__ __forceinline__ void arith_level_body(uint4 *__restrict__ W, uint64_t Bp, uint32_t B, const uint32_t *__restrict__ gate_stream,
                                                 const uint32_t *__restrict__ gate_offset, const uint32_t *__restrict__ consts,
                                                 uint32_t *__restrict__ event, const uint4 *__restrict__ Inv, uint32_t gate_index) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= B) return;
    GateWords g = GATE_WORDS(gate_stream) + gate_offset[gate_index];
    const GateDeviceLoader ld{W, Inv, consts, Bp, j};
    Fr29 local = fr29_from(fr_zero());
    bool host = true;
    for (;;) {  // the record, then the records fused behind it (they read this one's output as GATE_LOCAL)
        const uint32_t w0 = g[0], kind = w0 & 0xff, opcode = g[1], out = g[2];
        const Fr29 acc = gate_eval(ld, g, local);
        if (kind == 0) {  // constraint only (arithmetic.rs:92-102): the value is canonical
            uint32_t z = 0;
#pragma unroll
            for (int i = 0; i < 9; i++) z |= acc.v[i];
            if (z) flag_instance(event, j, opcode);
        } else {  // coefficients were pre-multiplied by -1/coeff on the host (arithmetic.rs:120)
#ifdef GATE_EXP_NO_STORES  // measurement only: the result leaves through one word (kept alive, never true)
            if (fr29_pack(acc).v[3] == 0x12345678u && acc.v[5] == 77u) event[j] = 0;
#else
            fr_store_nt(W, out, Bp, j, fr29_pack(acc));  // read again levels later, long after it left the caches: 5.38 -> 5.50 M witnesses/s
#endif
        }
        if (!(w0 & GATE_TAIL_FLAG)) break;
        if (host || (w0 & GATE_SETLOCAL_FLAG)) local = acc;  // the tails read the host's output until a record takes `local` over
        host = false;
        g += gate_record_words(g);
    }
}
__device__ __forceinline__ Fr coef_value(uint32_t coef, const uint32_t *__restrict__ consts) {
    if (coef == K_COEF_ONE) return fr_one();
    if (coef == K_COEF_MINUS_ONE) return fr_neg(fr_one());
    if (coef == K_COEF_ZERO) return fr_zero();
    return fr_const(consts, coef);
}
// R^2 mod p: to_montgomery(x) = mont_mul(x, R2)
__device__ __forceinline__ Fr fr_r2() {
    Fr r = {{0x45b69bd4u, 0x38c2e14bu, 0x85883377u, 0x0ffedb18u, 0xabc6e54du, 0x7840f9f0u, 0x848b0f05u, 0x0a054a3eu}};  // 2^522 mod p
    return r;
}
// Montgomery <-> canonical little-endian 8x32 integers
__device__ __forceinline__ Fr fr_to_canonical(const Fr &a) {
    Fr one = fr_zero();
    one.v[0] = 1;
    return fr_mul(a, one);
}
__device__ __forceinline__ Fr fr_from_canonical(const Fr &c) { return fr_mul(c, fr_r2()); }  // c < p
__device__ __forceinline__ Fr fr_from_u32(uint32_t x) {
    Fr c = fr_zero();
    c.v[0] = x;
    return fr_from_canonical(c);
}
// Montgomery form of a byte from a 8 KiB table (digest bytes and radix digits become one field element each): a 32-byte
// gather instead of a Montgomery product
static __constant__ uint32_t BYTE_MONT[256][8] = {
#include "byte_mont_table.inc"
};
__device__ __forceinline__ Fr fr_from_byte(uint32_t d) {
    const uint4 *p = (const uint4 *)BYTE_MONT[d & 0xffu];
    const uint4 lo = p[0], hi = p[1];
    return Fr{{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w}};
}
// The inverse direction (tools/gen_byte_table.py --key): the low 10 bits of the 256 forms are pairwise distinct, so they name the only
// byte a stored value can be; the value IS that byte exactly when the stored form equals that byte's table entry (the representation
// is a bijection). Byte-sized witnesses -- hash inputs, byte RANGE checks -- are recognised by two small gathers and a compare instead
// of the low limb of a Montgomery reduction (44 quarter-rate multiply-adds); anything else takes the reduction.
static __constant__ uint8_t BYTE_KEY[1024] = {
#include "byte_key_table.inc"
};
// The same Montgomery form by arithmetic: d * r - k * p with r = 2^261 mod p and k = floor(d r / p) = (2333 d) >> 13 for d < 256 (checked
// over all 256 values by tools/gen_byte_table.py --check). A wave-wide gather from the 8 KiB table touches up to 64 cache lines and is
// served by the CU's one vector L1 at about a line per cycle; with sixteen waves of a hash launch per CU doing nothing else that rate,
// not the ALUs and not HBM, bounded the launch (cycle counters around the phases of hash_coop_level_kernel: 82 k cycles to fetch and
// recognise 16 bytes per wave, 46 k to convert and store 8 outputs, 26 k for the hash itself). ~64 instructions, no memory.
__device__ __forceinline__ Fr fr_mont_of_byte(uint32_t d) {
    constexpr uint32_t R261[8] = {0x8fffff57u, 0x2fd4e156u, 0xa494b01au, 0x75bba827u, 0x819caa80u, 0x5301fa84u, 0x563d4475u, 0x0dc83629u};
    constexpr uint32_t PP[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    d &= 0xffu;
    const uint32_t k = (d * 2333u) >> 13;
    Fr out;
    int64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int64_t acc = carry + (int64_t)((uint64_t)d * R261[i]) - (int64_t)((uint64_t)k * PP[i]);
        out.v[i] = (uint32_t)acc;
        carry = acc >> 32;
    }
    return out;
}
__device__ __forceinline__ bool fr_is_byte(const Fr &a, uint32_t &d) {
    d = BYTE_KEY[a.v[0] & 1023u];
    return fr_eq(a, fr_mont_of_byte(d));
}
// low 29 bits of the canonical value of a reduced stored form (exact for every input: bytes by the tables, the rest by fr29_redc_low)
__device__ __forceinline__ uint32_t fr_low_limb(const Fr &a, bool &is_byte) {
    uint32_t d;
    is_byte = fr_is_byte(a, d);
    if (!is_byte) d = fr29_redc_low(fr29_from(a));
    return d;
}
// to_u8_vec (acvm/src/pwg/blackbox/signature/mod.rs:5-18: the last big-endian byte of each witness) of 32 witnesses, read as ONE big-endian
// 256-bit integer (limbs little-endian): a signature half, a public-key coordinate, a hashed message. Four rows are in flight per lane and
// the byte comes from the table path (fr_low_limb), where one row at a time with a Montgomery reduction apiece (160 per ECDSA opcode, 74 per
// SchnorrVerify) left the one wave a SIMD holds waiting for every row in turn.
template <class P>
static __device__ __noinline__ Fr load_be32_bytes(const P &p, const uint32_t *__restrict__ ws) {
    Fr r = fr_zero();
    for (uint32_t g = 0; g < 8u; g++) {
        const Fr a0 = p.load(ws[4u * g]), a1 = p.load(ws[4u * g + 1u]), a2 = p.load(ws[4u * g + 2u]), a3 = p.load(ws[4u * g + 3u]);
        bool b;
        const uint32_t limb = (fr_low_limb(a0, b) & 0xffu) << 24 | (fr_low_limb(a1, b) & 0xffu) << 16 | (fr_low_limb(a2, b) & 0xffu) << 8 | (fr_low_limb(a3, b) & 0xffu);
#pragma unroll
        for (int k = 0; k < 8; k++)
            if ((uint32_t)k == 7u - g) r.v[k] = limb;
    }
    return r;
}
// the same for a byte string of any length, four rows at a time: put(i, byte)
template <class P, class Put>
__device__ __forceinline__ void load_bytes(const P &p, const uint32_t *__restrict__ ws, uint32_t n, Put put) {
    uint32_t i = 0;
    for (; i + 4u <= n; i += 4u) {
        const Fr a0 = p.load(ws[i]), a1 = p.load(ws[i + 1u]), a2 = p.load(ws[i + 2u]), a3 = p.load(ws[i + 3u]);
        bool b;
        put(i, fr_low_limb(a0, b) & 0xffu);
        put(i + 1u, fr_low_limb(a1, b) & 0xffu);
        put(i + 2u, fr_low_limb(a2, b) & 0xffu);
        put(i + 3u, fr_low_limb(a3, b) & 0xffu);
    }
    for (; i < n; i++) {
        bool b;
        put(i, fr_low_limb(p.load(ws[i]), b) & 0xffu);
    }
}
// num_bits of a canonical integer (generic_ark.rs:214-221)
__device__ __forceinline__ uint32_t canon_num_bits(const Fr &c) {
    uint32_t n = 0;
#pragma unroll
    for (int i = 0; i < 8; i++)
        if (c.v[i]) n = 32u * i + (32u - __clz(c.v[i]));
    return n;
}
// reduce an arbitrary 256-bit integer below p (2^256 / p < 6)
__device__ __forceinline__ Fr canon_reduce(Fr x) {
    for (int it = 0; it < 5; it++) {
        Fr d;
        if (!fr_sub256(d, x, fr_modulus())) x = d;
    }
    return x;
}

// ------------------------------------------------------------------------------------------------ policies
struct FastPolicy {
    uint4 *W;
    uint64_t Bp, j;
    const uint32_t *__restrict__ slot_of = nullptr;  // witness -> row of the table (plan.cpp slot reuse); null: the witness index
    uint64_t sBp = 0, sj = 0;  // stride and lane of the record's per-lane scratch (ops_kernel.hpp: per wave, stride 64); 0: the table's
    static constexpr bool exact = false;
    __device__ __forceinline__ uint64_t scratch_stride() const { return sBp ? sBp : Bp; }
    __device__ __forceinline__ uint64_t scratch_lane() const { return sBp ? sj : j; }
    __device__ __forceinline__ uint32_t row(uint32_t w) const { return slot_of ? slot_of[w] : w; }
    __device__ __forceinline__ bool known(uint32_t) const { return true; }
    __device__ __forceinline__ Fr load(uint32_t w) const { return fr_load(W, row(w), Bp, j); }  // (nontemporal here: config-5 mix 30.4 -> 30.8 ms, not taken)
    // insert_value (pwg/mod.rs:338-357). `was_assigned` is the planner's static knowledge. Returns false on conflict.
    __device__ __forceinline__ bool insert(uint32_t w, const Fr &v, uint32_t was_assigned) const {
        if (was_assigned) return fr_eq(fr_load(W, row(w), Bp, j), v);  // never overwrite: the exact kernel needs the old value
        fr_store_nt(W, row(w), Bp, j, v);  // outputs of the level kernels are read by later launches (config 3: 0.447 -> 0.505 of the HBM roofline)
        return true;
    }
};
struct ExactPolicy {
    uint4 *W;
    uint64_t Bp, j;
    uint32_t *assigned;
    uint32_t n_slow, t;
    static constexpr bool exact = true;
    __device__ __forceinline__ uint64_t scratch_stride() const { return Bp; }  // (the exact lanes' scratch: [word][lane])
    __device__ __forceinline__ uint64_t scratch_lane() const { return j; }
    __device__ __forceinline__ bool known(uint32_t w) const { return (assigned[(uint64_t)(w >> 5) * n_slow + t] >> (w & 31)) & 1u; }
    __device__ __forceinline__ Fr load(uint32_t w) const { return fr_load(W, w, Bp, j); }
    __device__ __forceinline__ bool insert(uint32_t w, const Fr &v, uint32_t) const {
        const bool had = known(w);
        Fr old = fr_zero();
        if (had) old = fr_load(W, w, Bp, j);
        fr_store(W, w, Bp, j, v);  // the map holds the new value even on conflict
        assigned[(uint64_t)(w >> 5) * n_slow + t] |= 1u << (w & 31);
        return !had || fr_eq(old, v);
    }
};

// ------------------------------------------------------------------------------------------------ expressions
// Expression record: [n_mul, n_lin, qc, (coef, l, r) x n_mul, (coef, -1/coef, w) x n_lin]. Returns the record length.
__device__ __forceinline__ uint32_t expr_len(const uint32_t *e) { return 3u + 3u * e[0] + 3u * e[1]; }

// get_value (pwg/mod.rs:321-332): the expression must evaluate to a constant, else MissingAssignment(w) with w chosen
// like any_witness_from_expression (:362-372) on the partially evaluated expression.
template <class P>
__device__ __forceinline__ OpResult expr_value(const P &p, const uint32_t *__restrict__ e, const uint32_t *__restrict__ consts, Fr &out) {
    const uint32_t n_mul = e[0], n_lin = e[1], qc = e[2];
    Fr acc = qc == K_COEF_ZERO ? fr_zero() : fr_const(consts, qc);
    uint32_t first_lin = K_NONE, first_mul = K_NONE;
    const uint32_t *t = e + 3;
    for (uint32_t i = 0; i < n_mul; i++, t += 3) {
        const uint32_t coef = t[0], l = t[1], r = t[2];
        const bool kl = p.known(l), kr = p.known(r);
        if (kl && kr) {
            if (coef != K_COEF_ZERO) acc = fr_add(acc, apply_coef(fr_mul(p.load(l), p.load(r)), coef, consts));
        } else if (P::exact) {
            if (!kl && !kr) {
                if (coef != K_COEF_ZERO && first_mul == K_NONE) first_mul = l;
            } else if (coef != K_COEF_ZERO) {
                Fr v = apply_coef(p.load(kl ? l : r), coef, consts);
                if (!fr_is_zero(v) && first_lin == K_NONE) first_lin = kl ? r : l;
            }
        }
    }
    for (uint32_t i = 0; i < n_lin; i++, t += 3) {
        const uint32_t coef = t[0], w = t[2];
        if (p.known(w)) {
            if (coef != K_COEF_ZERO) acc = fr_add(acc, apply_coef(p.load(w), coef, consts));
        } else if (coef != K_COEF_ZERO && first_lin == K_NONE) first_lin = w;
    }
    out = acc;
    if (P::exact) {
        if (first_lin != K_NONE) return op_fail(DE_MISSING_ASSIGNMENT, first_lin);
        if (first_mul != K_NONE) return op_fail(DE_MISSING_ASSIGNMENT, first_mul);
    }
    return op_ok();
}

}  // namespace acvm
