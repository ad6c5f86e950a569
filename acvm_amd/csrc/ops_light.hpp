// ops_light.hpp -- device routines of the cheap non-arithmetic opcodes, written once for both policies:
//   RANGE                  acvm/src/pwg/blackbox/range.rs:7-18
//   AND / XOR              acvm/src/pwg/blackbox/logic.rs:11-56, acir_field/src/generic_ark.rs:328-355,446-473
//   RecursiveAggregation   acvm/src/pwg/blackbox/mod.rs:154-161 (outputs := 0)
//   Directive::Quotient    acvm/src/pwg/directives/mod.rs:28-59
//   Directive::ToLeRadix   acvm/src/pwg/directives/mod.rs:60-87
//   MemoryInit / MemoryOp  acvm/src/pwg/memory_op.rs:16-124
// Record layouts are produced by plan.cpp (emit_* functions) and documented there.
#pragma once
#include "ops_common.hpp"

namespace acvm {

// black-box pre-check (blackbox/mod.rs:55-62): first unassigned input in get_inputs_vec order
template <class P>
__device__ __forceinline__ OpResult bb_inputs_assigned(const P &p, const uint32_t *ws, uint32_t n, uint32_t stride) {
    if (P::exact)
        for (uint32_t i = 0; i < n; i++)
            if (!p.known(ws[i * stride])) return op_fail(DE_MISSING_ASSIGNMENT, ws[i * stride]);
    return op_ok();
}

// [K_RANGE, opcode, w, num_bits]
template <class P>
__device__ __forceinline__ OpResult op_range(const P &p, const uint32_t *__restrict__ r) {
    OpResult pre = bb_inputs_assigned(p, r + 2, 1, 1);
    if (pre.err) return pre;
    const Fr a = p.load(r[2]);
    if (r[3] <= 8u) {
        // byte-sized ranges (the bulk of a hashing circuit): the value is below 2^bits <= 256 exactly when its stored Montgomery form is the
        // one of a byte below 2^bits (the representation is a bijection) -- two table gathers and a compare (ops_common.hpp fr_is_byte)
        uint32_t d;
        if (!fr_is_byte(a, d) || (d >> r[3]) != 0u) return op_fail(DE_UNSATISFIED);
        return op_ok();
    }
    if (canon_num_bits(fr_to_canonical(a)) > r[3]) return op_fail(DE_UNSATISFIED);
    return op_ok();
}

// [K_RANGE_MULTI, first opcode, n, (opcode, witness, num_bits) x n]: up to 8 RANGE opcodes of one level in one lane (level schedule only:
// plan.cpp merges them so that a wave pays its launch and its dependent record / row latencies once per eight checks, with four rows in
// flight). Fails with aux0 = the first (lowest) failing opcode, which record_level_kernel turns into the instance's event.
template <class P>
__device__ __forceinline__ OpResult op_range_multi(const P &p, const uint32_t *__restrict__ r) {
    const uint32_t n = r[2];
    const uint32_t *it = r + 3;
    uint32_t bad = 0xFFFFFFFFu;
    for (uint32_t i = 0; i < n; i += 4u) {
        Fr a[4];
#pragma unroll
        for (uint32_t k = 0; k < 4u; k++) a[k] = p.load(it[3u * (i + k < n ? i + k : i) + 1u]);
#pragma unroll
        for (uint32_t k = 0; k < 4u; k++) {
            if (i + k >= n) continue;
            const uint32_t bits = it[3u * (i + k) + 2u];
            bool ok;
            if (bits <= 8u) {
                uint32_t d;
                ok = fr_is_byte(a[k], d) && (d >> bits) == 0u;
            } else ok = canon_num_bits(fr_to_canonical(a[k])) <= bits;
            if (!ok) bad = min(bad, it[3u * (i + k)]);
        }
    }
    if (bad != 0xFFFFFFFFu) return op_fail(DE_UNSATISFIED, bad);
    return op_ok();
}

// mask_vector_le (generic_ark.rs:446-473) on a canonical integer: keep the low num_bits bits
__device__ __forceinline__ Fr canon_mask(const Fr &c, uint32_t num_bits) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t lo = 32u * i;
        r.v[i] = num_bits <= lo ? 0u : (num_bits >= lo + 32u ? c.v[i] : (c.v[i] & ((1u << (num_bits - lo)) - 1u)));
    }
    return r;
}

// [K_LOGIC, opcode, is_xor, lhs, rhs, bits_l, bits_r, out, out_flag]
template <class P>
__device__ __forceinline__ OpResult op_logic(const P &p, const uint32_t *__restrict__ r) {
    if (P::exact) {
        if (!p.known(r[3])) return op_fail(DE_MISSING_ASSIGNMENT, r[3]);
        if (!p.known(r[4])) return op_fail(DE_MISSING_ASSIGNMENT, r[4]);
    }
    if (r[5] != r[6]) return op_fail_msg(DE_PANIC, 0, DM_LOGIC_BITS);  // logic.rs:17-20 assert_eq!
    const Fr a = canon_mask(fr_to_canonical(p.load(r[3])), r[5]);
    const Fr b = canon_mask(fr_to_canonical(p.load(r[4])), r[5]);
    Fr c;
#pragma unroll
    for (int i = 0; i < 8; i++) c.v[i] = r[2] ? (a.v[i] ^ b.v[i]) : (a.v[i] & b.v[i]);
    // from_be_bytes_reduce: only matters when num_bits >= 254
    if (!p.insert(r[7], fr_from_canonical(canon_reduce(c)), r[8])) return op_fail(DE_UNSATISFIED);
    return op_ok();
}

// [K_ZERO_OUT, opcode, n_in, n_out, in ws..., (out, flag)...]
template <class P>
__device__ __forceinline__ OpResult op_zero_out(const P &p, const uint32_t *__restrict__ r) {
    OpResult pre = bb_inputs_assigned(p, r + 4, r[2], 1);
    if (pre.err) return pre;
    const uint32_t *o = r + 4 + r[2];
    for (uint32_t i = 0; i < r[3]; i++)
        if (!p.insert(o[2 * i], fr_zero(), o[2 * i + 1])) return op_fail(DE_UNSATISFIED);
    return op_ok();
}

// 256-bit unsigned division of canonical integers (num-bigint semantics); b != 0. Register-resident: a divisor below 2^32
// (the integer divisions of real circuits) takes eight 64-by-32 divisions, anything else a restoring division that starts
// at the dividend's top bit.
__device__ __forceinline__ void canon_divrem(const Fr &a, const Fr &b, Fr &q, Fr &rem) {
    q = fr_zero();
    rem = fr_zero();
    if ((b.v[1] | b.v[2] | b.v[3] | b.v[4] | b.v[5] | b.v[6] | b.v[7]) == 0u) {
        const uint32_t d = b.v[0];
        uint64_t r = 0;
#pragma unroll
        for (int k = 7; k >= 0; k--) {
            const uint64_t cur = r << 32 | a.v[k];
            const uint64_t qq = cur / d;  // r < d, so the quotient fits 32 bits
            q.v[k] = (uint32_t)qq;
            r = cur - qq * d;
        }
        rem.v[0] = (uint32_t)r;
        return;
    }
    // The first bits(b) - 1 steps of the restoring division only shift dividend bits into the remainder (it stays below
    // 2^(bits(b) - 1) <= b, nothing can be subtracted): start with those bits in place. Two full-width operands, the common
    // large-divisor case, then take 1-2 steps instead of 254.
    const int na = (int)canon_num_bits(a), nb = (int)canon_num_bits(b);
    if (na < nb) {  // a < b
        rem = a;
        return;
    }
    const uint32_t s = (uint32_t)(na - (nb - 1));  // bits of a still to be processed, 1 <= s <= 256
    {   // rem = a >> s (s = 256 only for b = 1, which took the small-divisor path above)
        const uint32_t ws = s >> 5, bs = s & 31u;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            uint32_t lo = 0, hi = 0;
#pragma unroll
            for (int m = 0; m < 8; m++) {
                if ((uint32_t)m == (uint32_t)k + ws) lo = a.v[m];
                if ((uint32_t)m == (uint32_t)k + ws + 1u) hi = a.v[m];
            }
            rem.v[k] = bs ? (lo >> bs | hi << (32u - bs)) : lo;
        }
    }
    for (int i = (int)s - 1; i >= 0; i--) {
        const uint32_t limb = (uint32_t)i >> 5, sh = (uint32_t)i & 31u;
        uint32_t abit = 0;
#pragma unroll
        for (int k = 0; k < 8; k++)
            if ((uint32_t)k == limb) abit = (a.v[k] >> sh) & 1u;
#pragma unroll
        for (int k = 7; k > 0; k--) rem.v[k] = rem.v[k] << 1 | rem.v[k - 1] >> 31;
        rem.v[0] = rem.v[0] << 1 | abit;
        Fr dd;
        const uint32_t borrow = fr_sub256(dd, rem, b);
        const uint32_t bit = borrow ? 0u : 1u << sh;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            rem.v[k] = borrow ? rem.v[k] : dd.v[k];
            if ((uint32_t)k == limb) q.v[k] |= bit;
        }
    }
}

// [K_QUOTIENT, opcode, q, fq, r, fr, has_pred, E(a), E(b), E(pred)?]
template <class P>
__device__ __forceinline__ OpResult op_quotient(const P &p, const uint32_t *__restrict__ r, const uint32_t *__restrict__ consts) {
    const uint32_t *ea = r + 7, *eb = ea + expr_len(ea), *ep = eb + expr_len(eb);
    Fr va, vb, pred = fr_one();
    OpResult e = expr_value(p, ea, consts, va);
    if (e.err) return e;
    e = expr_value(p, eb, consts, vb);
    if (e.err) return e;
    if (r[6]) {
        e = expr_value(p, ep, consts, pred);
        if (e.err) return e;
    }
    Fr q = fr_zero(), rem = fr_zero();
    if (!fr_is_zero(pred) && !fr_is_zero(vb)) canon_divrem(fr_to_canonical(va), fr_to_canonical(vb), q, rem);
    if (!p.insert(r[2], fr_from_canonical(q), r[3])) return op_fail(DE_UNSATISFIED);  // q, r <= a < p
    if (!p.insert(r[4], fr_from_canonical(rem), r[5])) return op_fail(DE_UNSATISFIED);
    return op_ok();
}

// [K_TO_LE_RADIX, opcode, radix, n_out, (out, flag) x n_out, E(a)]
template <class P>
__device__ __forceinline__ OpResult op_to_le_radix(const P &p, const uint32_t *__restrict__ r, const uint32_t *__restrict__ consts) {
    const uint32_t radix = r[2], n_out = r[3];
    const uint32_t *outs = r + 4, *ea = outs + 2 * n_out;
    Fr va;
    OpResult e = expr_value(p, ea, consts, va);
    if (e.err) return e;
    if (radix < 2 || radix > 256) return op_fail_msg(DE_PANIC, 0, DM_RADIX);  // num-bigint to_radix_le assert
    Fr v = fr_to_canonical(va);
    // BigUint::to_radix_le: little-endian digits, 0 -> [0]. Digits are produced one by one; more digits than
    // outputs -> UnsatisfiedConstrain before anything is inserted (directives/mod.rs:67-71), so count first.
    uint32_t nd;
    const bool pow2 = (radix & (radix - 1)) == 0;
    const uint32_t log2r = 31u - __clz(radix);
    if (pow2) {
        const uint32_t nb = canon_num_bits(v);
        nd = nb == 0 ? 1u : (nb + log2r - 1) / log2r;
    } else {
        Fr t = v;
        nd = 0;
        do {
            uint64_t rem = 0;
#pragma unroll
            for (int k = 7; k >= 0; k--) {
                const uint64_t cur = rem << 32 | t.v[k];
                t.v[k] = (uint32_t)(cur / radix);
                rem = cur % radix;
            }
            nd++;
        } while (!fr_is_zero(t));
    }
    if (n_out < nd) return op_fail(DE_UNSATISFIED);
    Fr t = v;
    for (uint32_t i = 0; i < n_out; i++) {
        uint32_t digit = 0;
        if (i < nd) {
            if (pow2) {
                const uint32_t pos = i * log2r;  // a digit never straddles more than two limbs (log2r <= 8)
                uint32_t lo = 0, hi = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    if ((uint32_t)k == (pos >> 5)) lo = v.v[k];
                    if ((uint32_t)k == (pos >> 5) + 1) hi = v.v[k];
                }
                const uint64_t two = (uint64_t)hi << 32 | lo;
                digit = (uint32_t)(two >> (pos & 31)) & (radix - 1);
            } else {
                uint64_t rem = 0;
#pragma unroll
                for (int k = 7; k >= 0; k--) {
                    const uint64_t cur = rem << 32 | t.v[k];
                    t.v[k] = (uint32_t)(cur / radix);
                    rem = cur % radix;
                }
                digit = (uint32_t)rem;
            }
        }
        if (!p.insert(outs[2 * i], fr_from_byte(digit), outs[2 * i + 1])) return op_fail(DE_UNSATISFIED);
    }
    return op_ok();
}

// ------------------------------------------------------------------------------------------------ memory
// Per-instance memory blocks live in a table laid out like W: Mem[cell][half][instance].
// [K_MEM_INIT, opcode, cell_base, len, ws...]   (memory_op.rs:47-60)
template <class P>
__device__ __forceinline__ OpResult op_mem_init(const P &p, const uint32_t *__restrict__ r, uint4 *Mem) {
    const uint32_t base = r[2], len = r[3];
    for (uint32_t i = 0; i < len; i++) {
        const uint32_t w = r[4 + i];
        if (P::exact && !p.known(w)) return op_fail(DE_MISSING_ASSIGNMENT, w);
        fr_store(Mem, base + i, p.Bp, p.j, p.load(w));
    }
    return op_ok();
}

// partial evaluation summary of an expression (arithmetic.rs:212-239) for MemoryOp's value operand
struct ExprPartial {
    Fr constant, lin_coef;
    uint32_t n_lin, n_mul, lin_w, first_w;
};
template <class P>
__device__ __forceinline__ ExprPartial expr_partial(const P &p, const uint32_t *__restrict__ e, const uint32_t *__restrict__ consts) {
    ExprPartial out;
    const uint32_t n_mul = e[0], n_lin = e[1], qc = e[2];
    out.constant = qc == K_COEF_ZERO ? fr_zero() : fr_const(consts, qc);
    out.lin_coef = fr_zero();
    out.n_lin = out.n_mul = 0;
    out.lin_w = K_NONE;
    uint32_t first_lin = K_NONE, first_mul = K_NONE;
    const uint32_t *t = e + 3;
    for (uint32_t i = 0; i < n_mul; i++, t += 3) {
        const uint32_t coef = t[0], l = t[1], r = t[2];
        const bool kl = p.known(l), kr = p.known(r);
        if (kl && kr) {
            if (coef != K_COEF_ZERO) out.constant = fr_add(out.constant, apply_coef(fr_mul(p.load(l), p.load(r)), coef, consts));
        } else if (!kl && !kr) {
            if (coef != K_COEF_ZERO) { out.n_mul++; if (first_mul == K_NONE) first_mul = l; }
        } else if (coef != K_COEF_ZERO) {
            Fr v = apply_coef(p.load(kl ? l : r), coef, consts);
            if (!fr_is_zero(v)) {
                out.n_lin++; out.lin_coef = v; out.lin_w = kl ? r : l;
                if (first_lin == K_NONE) first_lin = out.lin_w;
            }
        }
    }
    for (uint32_t i = 0; i < n_lin; i++, t += 3) {
        const uint32_t coef = t[0], w = t[2];
        if (p.known(w)) {
            if (coef != K_COEF_ZERO) out.constant = fr_add(out.constant, apply_coef(p.load(w), coef, consts));
        } else if (coef != K_COEF_ZERO) {
            out.n_lin++; out.lin_coef = coef_value(coef, consts); out.lin_w = w;
            if (first_lin == K_NONE) first_lin = w;
        }
    }
    out.first_w = first_lin != K_NONE ? first_lin : first_mul;
    return out;
}

__device__ __forceinline__ bool canon_fits_u64(const Fr &c) { return (c.v[2] | c.v[3] | c.v[4] | c.v[5] | c.v[6] | c.v[7]) == 0; }

// [K_MEM_OP, opcode, cell_base, block_len, readable_len, has_pred, mode, target_w, target_flag,
//  E(operation), E(index), E(value), E(pred)?]   mode: 0 write, 1 read (planner: operation is a constant), 2 dynamic
// `replay` (exact path only, FastPolicy semantics): re-apply the memory side effect of an opcode that ran before the
// instance's event; the read target already holds the right value and no error is possible there.
template <class P>
__device__ __forceinline__ OpResult op_mem_op(const P &p, const uint32_t *__restrict__ r, const uint32_t *__restrict__ consts, uint4 *Mem,
                                              bool replay = false) {
    const uint32_t base = r[2], block_len = r[3], readable_len = r[4], has_pred = r[5];
    const uint32_t *e_op = r + 9, *e_idx = e_op + expr_len(e_op), *e_val = e_idx + expr_len(e_idx), *e_pred = e_val + expr_len(e_val);
    Fr operation, index, pred = fr_one();
    OpResult e = expr_value(p, e_op, consts, operation);
    if (e.err) return e;
    e = expr_value(p, e_idx, consts, index);
    if (e.err) return e;
    const Fr ci = fr_to_canonical(index);
    if (!canon_fits_u64(ci)) return op_fail_msg(DE_PANIC, 0, DM_MEM_INDEX_U64);  // try_to_u64().unwrap() (memory_op.rs:72)
    const uint32_t mi = ci.v[0];                                              // `as MemoryIndex` wraps to u32
    const bool is_read = fr_is_zero(operation);
    if (P::exact) {
        const ExprPartial v = expr_partial(p, e_val, consts);
        if (has_pred) {
            e = expr_value(p, e_pred, consts, pred);
            if (e.err) return e;
        }
        if (is_read) {
            // Expression::to_witness (expression/mod.rs:158-172)
            if (!(v.n_mul == 0 && v.n_lin == 1 && fr_eq(v.lin_coef, fr_one()) && fr_is_zero(v.constant)))
                return op_fail_msg(DE_PANIC, 0, DM_MEM_READ_EXPR);
            Fr val = fr_zero();
            if (!fr_is_zero(pred)) {
                if (mi >= readable_len) return op_fail(DE_INDEX_OOB, mi, block_len);  // key absent (memory_op.rs:37-44)
                val = fr_load(Mem, base + mi, p.Bp, p.j);
            }
            if (!p.insert(v.lin_w, val, 0)) return op_fail(DE_UNSATISFIED);
        } else if (!fr_is_zero(pred)) {
            if (v.n_mul || v.n_lin) return op_fail(DE_MISSING_ASSIGNMENT, v.first_w);  // get_value(&value_write)
            if (mi >= block_len) return op_fail(DE_INDEX_OOB, mi, block_len);
            fr_store(Mem, base + mi, p.Bp, p.j, v.constant);
        }
        return op_ok();
    }
    // generic instance: the planner fixed read / write and the read target
    if (has_pred) {
        e = expr_value(p, e_pred, consts, pred);
        if (e.err) return e;
    }
    if (r[6] == 1) {
        if (replay) return op_ok();
        Fr val = fr_zero();
        if (!fr_is_zero(pred)) {
            if (mi >= readable_len) return op_fail(DE_INDEX_OOB, mi, block_len);
            val = fr_load(Mem, base + mi, p.Bp, p.j);
        }
        if (!p.insert(r[7], val, r[8])) return op_fail(DE_UNSATISFIED);
    } else if (!fr_is_zero(pred)) {
        Fr val;
        e = expr_value(p, e_val, consts, val);
        if (e.err) return e;
        if (mi >= block_len) return op_fail(DE_INDEX_OOB, mi, block_len);
        fr_store(Mem, base + mi, p.Bp, p.j, val);
    }
    return op_ok();
}

// ------------------------------------------------------------------------------------------------ arithmetic (exact)
// ArithmeticSolver::solve (arithmetic.rs:27-127) with evaluate (:212-239), solve_mul_term (:133-144),
// solve_fan_in_term (:176-209) and insert_value (pwg/mod.rs:338-357) restated on the per-instance assigned set.
// [K_ARITH, opcode, E(expr)]. The level kernels use the planner's folded gate stream instead.
template <class P>
__device__ __forceinline__ OpResult op_arith(const P &p, const uint32_t *__restrict__ r, const uint32_t *__restrict__ consts) {
    const uint32_t *__restrict__ e = r + 2;
    const uint32_t n_mul = e[0], n_lin = e[1], qc = e[2];
    Fr acc = qc == K_COEF_ZERO ? fr_zero() : fr_const(consts, qc);
    uint32_t residual_mul = 0, unknowns = 0, unk_w = 0, unk_ninv = 0;
    bool unk_dynamic = false;
    Fr unk_c = fr_zero();
    const uint32_t *__restrict__ t = e + 3;
    for (uint32_t i = 0; i < n_mul; i++, t += 3) {
        const uint32_t coef = t[0], l = t[1], rr = t[2];
        const bool kl = p.known(l), kr = p.known(rr);
        if (kl && kr) {
            if (coef != K_COEF_ZERO) acc = fr_add(acc, apply_coef(fr_mul(p.load(l), p.load(rr)), coef, consts));
        } else if (!kl && !kr) {
            if (coef != K_COEF_ZERO) residual_mul++;
        } else if (coef != K_COEF_ZERO) {
            Fr v = apply_coef(p.load(kl ? l : rr), coef, consts);
            if (!fr_is_zero(v)) { unknowns++; unk_c = v; unk_w = kl ? rr : l; unk_dynamic = true; }
        }
    }
    for (uint32_t i = 0; i < n_lin; i++, t += 3) {
        const uint32_t coef = t[0], w = t[2];
        if (p.known(w)) {
            if (coef != K_COEF_ZERO) acc = fr_add(acc, apply_coef(p.load(w), coef, consts));
        } else if (coef != K_COEF_ZERO) {
            unknowns++;
            unk_ninv = t[1];
            unk_dynamic = false;
            unk_w = w;
        }
    }
    if (residual_mul >= 2) return op_fail_msg(DE_PANIC, 0, DM_TWO_MUL_TERMS);       // panic (arithmetic.rs:142)
    if (residual_mul == 1 || unknowns > 1) return op_fail(DE_TOO_MANY_UNKNOWNS);  // (:38-42)
    if (unknowns == 0) {
        if (!fr_is_zero(acc)) return op_fail(DE_UNSATISFIED);  // (:92-102)
        return op_ok();
    }
    // assignment = -(total_sum / coeff) (:86,120). Constant coefficients carry their -1/c from the planner; a coefficient
    // that is a product with a known witness (:217-221) is inverted per instance.
    const Fr val = unk_dynamic ? fr_neg(fr_mul(acc, fr_inv(unk_c))) : apply_coef(acc, unk_ninv, consts);
    if (!p.insert(unk_w, val, 0)) return op_fail(DE_UNSATISFIED);
    return op_ok();
}

// panic codes (host: brillig_panic_text in batch.cpp)
enum BrPanic : uint32_t {
    BP_REG_READ = 1, BP_REG_WRITE = 2, BP_U64 = 3, BP_MEM_READ = 4, BP_BITS_256 = 6, BP_SUB_OVERFLOW = 7, BP_DIV_ZERO = 8, BP_SHIFT_BITS = 9,
    BP_UNWRAP = 10, BP_BAD_INT_OP = 11, BP_BYTECODE_OOB = 12, BP_BAD_OPCODE = 13, BP_OUT_MEM_OOB = 15, BP_BAD_BB = 16
};

// ---- 256-bit helpers on canonical integers
__device__ __forceinline__ Fr int_mask(const Fr &a, uint32_t bits) { return canon_mask(a, bits); }
__device__ __forceinline__ int int_cmp(const Fr &a, const Fr &b) {
    Fr d;
    if (fr_sub256(d, a, b)) return -1;
    return fr_is_zero(d) ? 0 : 1;
}
__device__ __forceinline__ Fr int_pow2(uint32_t bits) {  // 2^bits, bits < 256
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = (bits >> 5) == (uint32_t)i ? 1u << (bits & 31u) : 0u;
    return r;
}
__device__ __forceinline__ Fr int_neg(const Fr &a) {
    Fr z = fr_zero(), r;
    fr_sub256(r, z, a);
    return r;
}
__device__ __forceinline__ Fr int_mul_lo(const Fr &a, const Fr &b) {  // low 256 bits of a * b
    uint32_t r[8];
#pragma unroll
    for (int i = 0; i < 8; i++) r[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int k = 0; k + i < 8; k++) {
            c += (uint64_t)a.v[i] * b.v[k] + r[i + k];
            r[i + k] = (uint32_t)c;
            c >>= 32;
        }
    }
    Fr o;
#pragma unroll
    for (int i = 0; i < 8; i++) o.v[i] = r[i];
    return o;
}
__device__ __forceinline__ Fr int_mul_full(const Fr &a, const Fr &b, Fr &hi) {  // a * b = hi 2^256 + (returned low half)
    uint32_t r[16];
#pragma unroll
    for (int i = 0; i < 16; i++) r[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            c += (uint64_t)a.v[i] * b.v[k] + r[i + k];
            r[i + k] = (uint32_t)c;
            c >>= 32;
        }
        r[i + 8] = (uint32_t)c;
    }
    Fr lo;
#pragma unroll
    for (int i = 0; i < 8; i++) { lo.v[i] = r[i]; hi.v[i] = r[8 + i]; }
    return lo;
}
__device__ __forceinline__ uint32_t limb_or_zero(const Fr &a, int idx) {
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < 8; k++)
        if (k == idx) r = a.v[k];
    return r;
}
__device__ __forceinline__ Fr int_shl(const Fr &a, uint32_t s) {  // s < 256
    const int q = (int)(s >> 5);
    const uint32_t rs = s & 31u;
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t v = limb_or_zero(a, i - q) << rs;
        if (rs) v |= limb_or_zero(a, i - q - 1) >> (32u - rs);
        r.v[i] = v;
    }
    return r;
}
__device__ __forceinline__ Fr int_shr(const Fr &a, uint32_t s) {  // s < 256
    const int q = (int)(s >> 5);
    const uint32_t rs = s & 31u;
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t v = limb_or_zero(a, i + q) >> rs;
        if (rs) v |= limb_or_zero(a, i + q + 1) << (32u - rs);
        r.v[i] = v;
    }
    return r;
}
// two's complement view of SignedDiv (arithmetic.rs:84-98): a < 2^(bits-1) -> (+, a); a < 2^bits -> (-, 2^bits - a); else (+, a - 2^bits)
__device__ __forceinline__ bool int_to_signed(const Fr &a, uint32_t bits, Fr &mag) {
    if (int_cmp(a, int_pow2(bits - 1u)) < 0) { mag = a; return false; }
    if (bits == 256u) { mag = int_neg(a); return true; }
    const Fr full = int_pow2(bits);
    if (int_cmp(a, full) < 0) { fr_sub256(mag, full, a); return true; }
    fr_sub256(mag, a, full);
    return false;
}

// evaluate_binary_bigint_op (arithmetic.rs:23-81) + the conversion back to a field element (from_be_bytes_reduce)
// `panic` receives a BrPanic code (and the result is meaningless) where the reference panics
// bit_size > 256 (the reference's BigUint takes any size; operands are field elements < p < 2^254): the masks are no-ops on 256-bit values,
// so only two ops still see the modulus 2^bits: Mul, whose 512-bit product is masked before it is reduced mod p, and Sub with a < b,
// whose result 2^bits + a - b survives only as its residue (2^bits mod p) + a - b -- `pow2` = the Montgomery form of 2^bits mod p, a
// constant the planner computes per instruction. SignedDiv sees two non-negative numbers. Shifts panic above 128 bits either way.
static inline __device__ Fr int_op_core(uint32_t op, uint32_t bits, const Fr &fa, const Fr &fb, uint32_t &panic, const Fr *pow2 = nullptr) {
    Fr a = fr_to_canonical(fa), b = fr_to_canonical(fb), r = fr_zero();
    switch (op) {
    case 0: fr_add256(r, a, b); r = int_mask(r, bits); break;  // a, b < 2^254: no carry out of 256 bits
    case 1: {  // (2^bits + a - b) % 2^bits; BigUint underflow when b > 2^bits + a
        const bool borrow = fr_sub256(r, a, b) != 0;
        if (borrow && bits > 256u) return pow2 ? fr_add(*pow2, fr_sub(fa, fb)) : (panic = BP_BITS_256, fr_zero());
        if (borrow && bits < 256u && int_cmp(int_neg(r), int_pow2(bits)) > 0) { panic = BP_SUB_OVERFLOW; return fr_zero(); }
        r = int_mask(r, bits);
        break;
    }
    case 2:
        if (bits > 256u) {  // (a b mod 2^bits) mod p = lo + hi 2^256 with the high half masked at bits - 256
            Fr hi;
            r = int_mul_full(a, b, hi);
            hi = int_mask(hi, bits - 256u);
            const Fr c256 = {{0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u, 0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u}};  // 2^256 mod p
            return fr_add(fr_from_canonical(canon_reduce(r)), fr_mul(fr_from_canonical(canon_reduce(hi)), fr_from_canonical(c256)));
        }
        r = int_mask(int_mul_lo(a, b), bits);
        break;
    case 3: {  // SignedDiv
        if (bits == 0u) { panic = BP_SUB_OVERFLOW; return r; }
        Fr ma, mb, q, rem;
        bool sa = false, sb = false;
        if (bits > 256u) { ma = a; mb = b; }  // a, b < 2^254 <= 2^(bits - 1): both non-negative
        else { sa = int_to_signed(a, bits, ma); sb = int_to_signed(b, bits, mb); }
        if (fr_is_zero(mb)) { panic = BP_DIV_ZERO; return r; }
        canon_divrem(ma, mb, q, rem);
        if (!((sa != sb) && !fr_is_zero(q))) r = q;
        else if (bits == 256u) r = int_neg(q);
        else {
            if (int_cmp(q, int_pow2(bits)) > 0) { panic = BP_SUB_OVERFLOW; return r; }
            fr_sub256(r, int_pow2(bits), q);
        }
        break;
    }
    case 4: {  // UnsignedDiv
        a = int_mask(a, bits);
        b = int_mask(b, bits);
        if (fr_is_zero(b)) { panic = BP_DIV_ZERO; return r; }
        Fr rem;
        canon_divrem(a, b, r, rem);
        break;
    }
    case 5: case 6: case 7: {
        const int c = int_cmp(int_mask(a, bits), int_mask(b, bits));
        r.v[0] = op == 5u ? c == 0 : (op == 6u ? c < 0 : c <= 0);
        break;
    }
    case 8: case 9: case 10:
#pragma unroll
        for (int i = 0; i < 8; i++) r.v[i] = op == 8u ? a.v[i] & b.v[i] : (op == 9u ? a.v[i] | b.v[i] : a.v[i] ^ b.v[i]);
        r = int_mask(r, bits);
        break;
    case 11: case 12: {
        if (bits > 128u) { panic = BP_SHIFT_BITS; return r; }
        if (b.v[4] | b.v[5] | b.v[6] | b.v[7]) { panic = BP_UNWRAP; return r; }  // to_u128().unwrap()
        const bool small = !(b.v[1] | b.v[2] | b.v[3]) && b.v[0] < 256u;
        if (small) r = int_mask(op == 11u ? int_shl(a, b.v[0]) : int_shr(a, b.v[0]), bits);
        break;
    }
    default: panic = BP_BAD_INT_OP; return r;
    }
    return fr_from_canonical(canon_reduce(r));
}


// ------------------------------------------------------------------------------------------------ straight-line Brillig
// [K_BRILLIG_SL, opcode, has_pred, n_inputs, n_outputs, n_ins, E(pred)?, E(input) x n_inputs, (w, flag) x n_outputs,
//  (kind | sub_op << 8 | dst << 16 | a << 20 | b << 24, bit_size or jump target, constant) x n_ins]
// Level schedule only (plan.cpp "straight-line Brillig"): an Opcode::Brillig whose bytecode is BinaryFieldOp / BinaryIntOp / Const / Mov /
// Stop / Trap and FORWARD jumps over at most BRILLIG_SL_REGS registers, with single-value inputs and outputs -- the stdlib's integer
// fallbacks (stdlib/src/blackbox_fallbacks/uint.rs:212-260), the inversion and comparison helpers the compiler emits -- runs here without
// the VM: same arithmetic (brillig_vm/src/arithmetic.rs:7-81), same register semantics (unset registers read 0, registers.rs:25-33;
// input i in register i, register i into output i, pwg/brillig.rs:46-111). The wave walks the instructions once, in order; a lane
// executes instruction i when its own program counter stands on it (a forward jump only skips ahead), so control flow stays
// wave-uniform. The registers live in LDS (reg r, word k of lane l at regs[(8 r + k) * BLOCK + l]). Anything the reference would report
// -- a panicking integer op, a trap, an output conflict -- only flags the instance; the exact path then runs the opcode's own
// K_BRILLIG record in the VM, which words the failure.
static constexpr uint32_t BRILLIG_SL_REGS = 4;
static constexpr uint32_t LIGHT_SL_BLOCK = 128;
enum SlKind : uint32_t { SL_FIELD = 0, SL_INT = 1, SL_CONST = 2, SL_MOV = 3, SL_JUMP = 4, SL_JUMP_IF = 5, SL_JUMP_IF_NOT = 6, SL_STOP = 7, SL_TRAP = 8 };
__device__ __forceinline__ Fr sl_get(const uint32_t *regs, uint32_t r) {
    Fr x;
#pragma unroll
    for (int k = 0; k < 8; k++) x.v[k] = regs[(8u * r + (uint32_t)k) * LIGHT_SL_BLOCK];
    return x;
}
__device__ __forceinline__ void sl_set(uint32_t *regs, uint32_t r, const Fr &x) {
#pragma unroll
    for (int k = 0; k < 8; k++) regs[(8u * r + (uint32_t)k) * LIGHT_SL_BLOCK] = x.v[k];
}
template <class P>
__device__ __forceinline__ OpResult op_brillig_sl(const P &p, const uint32_t *__restrict__ r, const uint32_t *__restrict__ consts, uint32_t *lds) {
    if (P::exact || !lds) return op_fail_msg(DE_PANIC, 0, DM_NONE);  // never scheduled on the exact path
    const uint32_t has_pred = r[2], n_inputs = r[3], n_outputs = r[4], n_ins = r[5];
    const uint32_t *q = r + 6;
    Fr pred = fr_one();
    if (has_pred) {
        const OpResult e = expr_value(p, q, consts, pred);
        if (e.err) return e;
        q += expr_len(q);
    }
    uint32_t *regs = lds + threadIdx.x;
    for (uint32_t i = 0; i < BRILLIG_SL_REGS; i++) sl_set(regs, i, fr_zero());
    for (uint32_t i = 0; i < n_inputs; i++) {
        Fr v;
        const OpResult e = expr_value(p, q, consts, v);
        if (e.err) return op_fail(DE_TOO_MANY_UNKNOWNS);
        q += expr_len(q);
        sl_set(regs, i, v);
    }
    const uint32_t *outs = q, *ins = outs + 2 * n_outputs;
    const bool skip = fr_is_zero(pred);  // zero_out_brillig_outputs (brillig.rs:133-150)
    uint32_t pc = skip ? 0xFFFFFFFFu : 0u;  // this lane's program counter; 0xFFFFFFFF = finished
    bool bad = false;
    for (uint32_t i = 0; i < n_ins; i++, ins += 3) {
        const bool act = pc == i;
        if (__builtin_amdgcn_ballot_w64(act) == 0) continue;  // (no lane of the wave stands here: jumped over, or all done)
        const uint32_t w = ins[0], kind = w & 0xffu, sub = (w >> 8) & 0xffu, dst = (w >> 16) & 0xfu, ra = (w >> 20) & 0xfu, rb = (w >> 24) & 0xfu;
        uint32_t next = i + 1u;
        if (kind <= SL_MOV) {
            Fr v;
            if (kind == SL_CONST) v = fr_const(consts, ins[2]);
            else if (kind == SL_MOV) v = sl_get(regs, ra);
            else {
                const Fr x = sl_get(regs, ra), y = sl_get(regs, rb);
                if (kind == SL_FIELD) {
                    switch (sub) {
                    case 0: v = fr_add(x, y); break;
                    case 1: v = fr_sub(x, y); break;
                    case 2: v = fr_mul(x, y); break;
                    case 3: v = fr_mul(x, fr_inv(y)); break;
                    default: v = fr_eq(x, y) ? fr_one() : fr_zero(); break;
                    }
                } else {
                    uint32_t panic = 0;
                    const Fr pow2 = ins[1] > 256u && sub == 1u ? fr_const(consts, ins[2]) : fr_zero();  // 2^bit_size mod p of a wide Sub (plan.cpp)
                    v = int_op_core(sub, ins[1], x, y, panic, &pow2);
                    if (panic && act) { bad = true; next = 0xFFFFFFFFu; }
                }
            }
            if (act) sl_set(regs, dst, v);
        } else if (kind == SL_JUMP) next = ins[1];
        else if (kind == SL_JUMP_IF || kind == SL_JUMP_IF_NOT) {
            const bool zero = fr_is_zero(sl_get(regs, ra));
            if (zero == (kind == SL_JUMP_IF_NOT)) next = ins[1];
        } else {  // Stop, Trap
            next = 0xFFFFFFFFu;
            if (kind == SL_TRAP && act) bad = true;
        }
        if (act) pc = next;
    }
    if (bad) return op_fail(DE_PANIC);
    bool ok = true;
    for (uint32_t i = 0; i < n_outputs; i++)
        ok = p.insert(outs[2 * i], skip ? fr_zero() : sl_get(regs, i), outs[2 * i + 1]) && ok;
    return ok ? op_ok() : op_fail(DE_UNSATISFIED);
}

// every record kind of class CLS_LIGHT
template <class P>
__device__ __forceinline__ OpResult dispatch_light(const P &p, const uint32_t *__restrict__ r, const uint32_t *__restrict__ consts, uint4 *Mem) {
    switch (r[0]) {
    case K_ARITH: return op_arith(p, r, consts);
    case K_RANGE: return op_range(p, r);
    case K_RANGE_MULTI: return op_range_multi(p, r);
    case K_LOGIC: return op_logic(p, r);
    case K_ZERO_OUT: return op_zero_out(p, r);
    case K_QUOTIENT: return op_quotient(p, r, consts);
    case K_TO_LE_RADIX: return op_to_le_radix(p, r, consts);
    case K_MEM_INIT: return op_mem_init(p, r, Mem);
    case K_MEM_OP: return op_mem_op(p, r, consts, Mem);
    default: return op_fail_msg(DE_PANIC, 0, DM_NONE);
    }
}

}  // namespace acvm
