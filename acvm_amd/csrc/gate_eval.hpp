// gate_eval.hpp -- the folded Arithmetic gate of the level kernels: ArithmeticSolver::solve (acvm/src/pwg/arithmetic.rs:27-127) for the generic
// instance of plan.cpp, one record = sum q_i a_i b_i + sum q_j w_j + q_c with the unknown's -1/coeff folded into every coefficient.
// __host__ __device__ and templated on the operand loader, so that the same code that arith_level_kernel runs is executed on the host
// against big integers with the planner's bounds checked (tools/gate_host_test.hip, tests/test_gate_eval_on_host.py).
//
// RELAXED ROWS (round 5). A witness that only Arithmetic gates read ("projective" in plan.cpp: it already carries a scale) is stored as
// ANY representative of its residue below 2^256 (5.29 p), not the canonical one: the Montgomery product tolerates such operands (fr_device.hpp),
// so the gate that writes it skips the canonicalisation -- a quotient-estimate reduction, a conditional subtraction and a ballot, 50-85 of
// the ~410 VALU instructions of a gate -- and the carry pass before it: the terms that are only added (the constant, +-1 linear terms, earlier
// dot products) ride in the upper columns of the gate's last Montgomery reduction, whose column scan hands out normalised limbs anyway.
// The planner tracks a bound per witness in units of p / 256 (plan.cpp `kbound`, mirrored from the walk below) and asks the record for
// a reduction only where the bound would pass 2^256 (GATE_OUT_WEAK), or for the canonical value where a consumer needs it (GATE_OUT_CANON:
// a witness any non-Arithmetic opcode, the inversion kernel or a caller may read as it is). Equality in the reference is on canonical
// values (acir_field/src/generic_ark.rs:88-92,164-169): every reader outside the gate kernels sees canonical values because the relaxed
// witnesses are exactly the scaled ones, whose readers multiply by 1 / scale first (export, digest, the exact path's hand-over).
#pragma once
#include "fr_device.hpp"
#include "gate_record.hpp"

namespace acvm {

FR_HD __forceinline__ void gate_h_room(Fr29 &h, uint32_t &hw, uint32_t weight) {
    if (hw + weight > GATE_H_MAX) {  // rare: many terms in one gate
        h = fr29_weak(fr29_norm(h));
        hw = GATE_H_AFTER_WEAK;
    }
    hw += weight;
}
// h += 2^klog2 p - x; x normalised, <= 2^klog2 p
FR_HD __forceinline__ void gate_h_sub(Fr29 &h, const Fr29 &x, uint32_t klog2) {
    if (klog2 <= 1) {
#pragma unroll
        for (int i = 0; i < 9; i++) h.v[i] += fr_kp29_sub(1, i) - x.v[i];
    } else if (klog2 == 2) {
#pragma unroll
        for (int i = 0; i < 9; i++) h.v[i] += fr_kp29_sub(2, i) - x.v[i];
    } else {
#pragma unroll
        for (int i = 0; i < 9; i++) h.v[i] += fr_kp29_sub(3, i) - x.v[i];
    }
}
// 8 inline words of a record -> working form (wave-uniform: scalar work on the device)
FR_HD __forceinline__ Fr29 gate_coef29(GateWords t) {
    Fr c;
#pragma unroll
    for (int i = 0; i < 8; i++) c.v[i] = t[i];
    return fr29_from(c);
}

// an operand row, or the wave's `local` registers (wave-uniform branch)
template <class L>
FR_HD __forceinline__ Fr29 gate_operand(const L &ld, uint32_t slot, const Fr29 &local) {
    if (slot == GATE_LOCAL) {
        Fr29 r = local;
#if defined(__HIP_DEVICE_COMPILE__)
        // (hipcc otherwise copies the forwarded value into the operand's registers AHEAD of this test -- nine moves in front of every operand fetch, taken
        // or not: a fortieth of the kernel's instructions; an empty volatile asm cannot be speculated, so the copy stays on the path that needs it)
#pragma unroll
        for (int k = 0; k < 9; k++) asm volatile("" : "+v"(r.v[k]));
#endif
        return r;
    }
    return ld.load(slot);
}
// the witness side of the k-th multiplied term of the record: the products come first (coef[8], a, b), then the linear terms (coef[8], w);
// its coefficient is gate_coef29 of the same entry (kept apart: the coefficient is wave-uniform and stays in scalar registers)
FR_HD __forceinline__ GateWords gate_mac_entry(GateWords t0, uint32_t np_mac, uint32_t k) {
    return k < np_mac ? t0 + 10 * k : t0 + 10 * np_mac + 9 * (k - np_mac);
}
template <class L>
FR_HD __forceinline__ Fr29 gate_mac_operand(const L &ld, GateWords c, bool product, const Fr29 &local) {
    if (product) return fr29_mul_b(gate_operand(ld, c[8], local), gate_operand(ld, c[9], local));
    return gate_operand(ld, c[8], local);
}

// Value of a record: normalised limbs, below the planner's bound for the record (plan.cpp mirrors this walk term by term).
//   L: loader with  Fr29 load(uint32_t slot) const  (GATE_LOCAL never reaches it),  Fr29 load_inverse(uint32_t slot) const,
//      GateWords constant(uint32_t idx) const  (8 words of the constant pool),  bool any(bool) const  (wave-wide OR: the ballot)
// Order: the terms that are only added or subtracted go into the lazy sum h first; then every Montgomery reduction of the record --
// two multiplied terms apiece -- takes the running sum along in its upper columns (fr29_dot_add), so the sum comes out of the last one
// with its carries propagated. An ASSERT record's value is canonical (zero test by the caller, arithmetic.rs:92-102).
template <class L>
FR_HD __forceinline__ Fr29 gate_eval(const L &ld, GateWords g, const Fr29 &local) {
    const uint32_t w0 = g[0], w5 = g[5], qc = g[3], kind = w0 & 0xff;
    const uint32_t np_mac = (w0 >> 8) & 0xff, nl_mac = (w0 >> 16) & 0xff, n_mac = np_mac + nl_mac;
    const uint32_t np_pos = w5 & 0xff, np_neg = (w5 >> 8) & 0xff, nl_pos = (w5 >> 16) & 0xff, nl_neg = w5 >> 24;
    const uint32_t sub_k = (w0 >> GATE_SUBK_SHIFT) & 3u;
    Fr29 h;
    uint32_t hw = 0;
    if (qc == GATE_COEF_ZERO) {
#pragma unroll
        for (int i = 0; i < 9; i++) h.v[i] = 0;
    } else {
        h = gate_coef29(ld.constant(qc));
        hw = 16;
    }
    GateWords t0 = g + 6;                          // np_mac x (coef[8], a, b), nl_mac x (coef[8], w)
    GateWords tp = t0 + 10 * np_mac + 9 * nl_mac;  // np_pos x (a, b)
    GateWords t = tp + 2 * np_pos;                 // np_neg x (a, b), nl_pos x (w), nl_neg x (w)
    // ---- the terms that are only added or subtracted
    for (uint32_t i = 0; i < np_neg; i++, t += 2) {
        const Fr29 x = fr29_mul_b(gate_operand(ld, t[0], local), gate_operand(ld, t[1], local));  // < 1.17 p
        gate_h_room(h, hw, 33);
        gate_h_sub(h, x, 1);
    }
    for (uint32_t i = 0; i < nl_pos; i++, t += 1) {
        const Fr29 x = gate_operand(ld, t[0], local);
        gate_h_room(h, hw, 16);
        h = fr29_addl(h, x);
    }
    for (uint32_t i = 0; i < nl_neg; i++, t += 1) {
        const Fr29 x = gate_operand(ld, t[0], local);
        gate_h_room(h, hw, 33);
        gate_h_sub(h, x, sub_k);
    }
    // ---- the multiplied terms. A product with coefficient +1 (the planner's projective witnesses make that the common product) shares its
    // reduction with a coefficient term: a b + c x is one fr29_dot<2>; then the coefficient terms two by two; then the +1 products that are left.
    uint32_t im = 0, ip = 0, n_red = 0;
    bool normalised = false;
    // (a record of hundreds of terms: the running sum is brought back below 1.03 p every GATE_REDUCTIONS_PER_WEAK reductions, long before
    // it could leave the range of fr29_weak -- wave-uniform, never taken by the gates of width-3 circuits)
#define GATE_AFTER_REDUCTION() do { normalised = true; if (++n_red == GATE_REDUCTIONS_PER_WEAK) { h = fr29_weak(h); n_red = 0; } } while (0)
    for (; ip < np_pos && im < n_mac; ip++, im++) {
        GateWords c = gate_mac_entry(t0, np_mac, im);
        const Fr29 l[2] = {gate_operand(ld, tp[2 * ip], local), gate_mac_operand(ld, c, im < np_mac, local)};
        const Fr29 m[2] = {gate_operand(ld, tp[2 * ip + 1], local), gate_coef29(c)};
        h = fr29_dot_add_b<2, 2u>(l, m, h);  // (fr_device.hpp: the asm-block scan; bit t of the mask = m[t] is a coefficient, wave-uniform)
        GATE_AFTER_REDUCTION();
    }
    for (; im < n_mac; im += 2) {
        GateWords c0 = gate_mac_entry(t0, np_mac, im);
        if (n_mac - im == 1) {
            const Fr29 l[1] = {gate_mac_operand(ld, c0, im < np_mac, local)}, m[1] = {gate_coef29(c0)};
            h = fr29_dot_add_b<1, 1u>(l, m, h);
        } else {
            GateWords c1 = gate_mac_entry(t0, np_mac, im + 1);
            const Fr29 l[2] = {gate_mac_operand(ld, c0, im < np_mac, local), gate_mac_operand(ld, c1, im + 1 < np_mac, local)};
            const Fr29 m[2] = {gate_coef29(c0), gate_coef29(c1)};
            h = fr29_dot_add_b<2, 3u>(l, m, h);
        }
        GATE_AFTER_REDUCTION();
    }
    for (; ip < np_pos; ip += 2) {
        if (np_pos - ip == 1) {
            const Fr29 l[1] = {gate_operand(ld, tp[2 * ip], local)}, m[1] = {gate_operand(ld, tp[2 * ip + 1], local)};
            h = fr29_dot_add_b<1, 0u>(l, m, h);
        } else {
            const Fr29 l[2] = {gate_operand(ld, tp[2 * ip], local), gate_operand(ld, tp[2 * ip + 2], local)};
            const Fr29 m[2] = {gate_operand(ld, tp[2 * ip + 1], local), gate_operand(ld, tp[2 * ip + 3], local)};
            h = fr29_dot_add_b<2, 0u>(l, m, h);
        }
        GATE_AFTER_REDUCTION();
    }
#undef GATE_AFTER_REDUCTION
    Fr29 acc = normalised ? h : fr29_norm(h);
    if (kind == 2) {
        // the unknown is multiplied by a known witness (arithmetic.rs:68-91): out = sum' / partner, and 1 / partner was put into the
        // inverse table by an earlier inverse_batch_kernel (a representative below 1.4 p)
        if (w0 & GATE_PRESUM_WEAK) acc = fr29_weak(acc);
        acc = fr29_mul_b(acc, ld.load_inverse(g[4]));
    }
    const uint32_t mode = kind == 0 ? GATE_OUT_CANON : (w0 >> GATE_OUT_SHIFT) & 3u;
    if (mode != GATE_OUT_ASIS) {
        acc = fr29_weak(acc);  // < 1.03 p
        if (mode == GATE_OUT_CANON) {
            // the step down from [p, 1.03 p) is taken only when some lane's top limb says it may be needed (p's top limb is reached by 2^-22
            // of the canonical values)
            if (ld.any(acc.v[8] >= fr_p29(8))) acc = fr29_csub(acc, 0);
        }
    }
    return acc;
}

}  // namespace acvm
