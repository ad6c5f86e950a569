// ops_sort.hpp -- Directive::PermutationSort on the device, one lane per (instance, directive):
//   acvm/src/pwg/directives/mod.rs:88-119     evaluate the tuples, stable sort by the `sort_by` columns (integer order of the
//                                             canonical values), control bits -> witnesses
//   acvm/src/pwg/directives/sorting.rs:8-244  SortingNetwork + route: control bits of the recursive permutation network that
//                                             maps the identity to the sorted order
// The reference keys its BTreeMaps by FieldElement; the values are the element indices, so the maps are arrays indexed by
// value and `free` (smallest element first) is a flag array scanned from 0. Everything lives in per-lane device scratch
// (word w of instance j at base[w * Bp + j]); the recursion is a depth-first walk with an explicit stack, emitting each
// node's switch_x, switch_y before its two sub-networks exactly like route()'s result.extend order. Sequential pointer
// chasing per lane: a rare directive, written for exactness.
#pragma once
#include "ops_common.hpp"

namespace acvm {

struct LaneWords {
    uint32_t *base;
    uint64_t Bp, j;
    __device__ __forceinline__ uint32_t get(uint32_t w) const { return base[(uint64_t)w * Bp + j]; }
    __device__ __forceinline__ void set(uint32_t w, uint32_t v) const { base[(uint64_t)w * Bp + j] = v; }
};

// [K_PERM_SORT, opcode, n, tuple, n_sort_by, n_bits, sort_by x n_sort_by, (bit witness, flag) x n_bits, E x (n * tuple)]
template <class P>
__device__ __forceinline__ OpResult op_perm_sort(const P &p, const uint32_t *__restrict__ r, const uint32_t *__restrict__ consts, uint32_t *scratch) {
    const uint32_t n = r[2], tuple = r[3], n_sort_by = r[4], n_bits = r[5];
    const uint32_t *sort_by = r + 6, *bit_ws = sort_by + n_sort_by, *e = bit_ws + 2 * n_bits;
    const LaneWords S{scratch, p.scratch_stride(), p.scratch_lane()};
    // scratch map (words): values [0, 8 n tuple) | order [n] | x_values [n] | y_values [n] | bits [..] | bump region
    const uint32_t o_vals = 0, o_order = 8 * n * tuple, o_xv = o_order + n, o_yv = o_xv + n, o_base = o_yv + n;
    // ---- evaluate (mod.rs:91-101): get_value of every tuple component, in order
    for (uint32_t i = 0; i < n * tuple; i++) {
        Fr v;
        const OpResult er = expr_value(p, e, consts, v);
        if (er.err) return er;
        e += expr_len(e);
        const Fr c = fr_to_canonical(v);
#pragma unroll
        for (int k = 0; k < 8; k++) S.set(o_vals + 8 * i + k, c.v[k]);
    }
    // ---- stable insertion sort of the indices by the sort_by columns (column `tuple` is the index itself)
    for (uint32_t i = 0; i < n; i++) S.set(o_order + i, i);
    for (uint32_t i = 1; i < n; i++) {
        const uint32_t cur = S.get(o_order + i);
        uint32_t jj = i;
        while (jj > 0) {
            const uint32_t prev = S.get(o_order + jj - 1);
            int cmp = 0;
            for (uint32_t s = 0; s < n_sort_by && !cmp; s++) {
                const uint32_t col = sort_by[s];
                if (col == tuple) { cmp = (prev > cur) - (prev < cur); continue; }
                // a[*i as usize] on a Vec of tuple + 1 elements (mod.rs:102-105): the comparator panics when it reaches the column
                if (col > tuple) return op_fail_msg(DE_PANIC, 0, DM_SORT_TUPLE, tuple + 1, col);
                for (int k = 7; k >= 0 && !cmp; k--) {
                    const uint32_t x = S.get(o_vals + 8 * (prev * tuple + col) + k), y = S.get(o_vals + 8 * (cur * tuple + col) + k);
                    cmp = (x > y) - (x < y);
                }
            }
            if (cmp <= 0) break;
            S.set(o_order + jj, prev);
            jj--;
        }
        S.set(o_order + jj, cur);
    }
    // ---- route(identity, order): depth-first over the sub-networks. A frame = (inputs offset, outputs offset, size).
    // The top-level inputs (identity) are materialised too so that every node reads its wires from scratch.
    uint32_t bump = o_base;
    const uint32_t o_bits = bump;
    uint32_t n_out = 0;
    {   // upper bound of the number of switches: n * ceil(log2 n)
        uint32_t lg = 0;
        while ((1u << lg) < n) lg++;
        bump += n * (lg + 1) + 1;
    }
    const uint32_t o_ident = bump;
    bump += n;
    for (uint32_t i = 0; i < n; i++) S.set(o_ident + i, i);
    const uint32_t o_stack = bump;  // 3 words per frame, depth <= 2 * 32
    bump += 3 * 72;
    uint32_t sp = 0;
    S.set(o_stack, o_ident); S.set(o_stack + 1, o_order); S.set(o_stack + 2, n);
    sp = 1;
    while (sp) {
        sp--;
        const uint32_t xin = S.get(o_stack + 3 * sp), yin = S.get(o_stack + 3 * sp + 1), m = S.get(o_stack + 3 * sp + 2);
        if (m <= 1) continue;
        if (m == 2) { S.set(o_bits + n_out++, S.get(xin) != S.get(yin) ? 1u : 0u); continue; }
        const uint32_t n1 = m / 2, nf = (m - 1) / 2;
        const uint32_t inner_x = bump, inner_y = bump + m, sw_x = bump + 2 * m, sw_y = sw_x + m / 2, fr_sw = sw_y + nf;
        bump = fr_sw + nf;
        for (uint32_t i = 0; i < m; i++) {
            S.set(o_xv + S.get(xin + i), i);
            S.set(o_yv + S.get(yin + i), i);
            S.set(inner_x + i, 0u);
            S.set(inner_y + i, 0u);
        }
        for (uint32_t i = 0; i < m / 2; i++) S.set(sw_x + i, 0u);
        for (uint32_t i = 0; i < nf; i++) { S.set(sw_y + i, 0u); S.set(fr_sw + i, 1u); }
        uint32_t n_free = nf;
        auto single_x = [&](uint32_t a) { return (m & 1u) && a == m - 1; };
        auto single_y = [&](uint32_t a) { return a >= m - 2 + (m & 1u); };
        auto inner_of = [&](uint32_t idx, uint32_t sw) { return (sw ^ (idx & 1u)) ? idx / 2 + m / 2 : idx / 2; };
        auto conf_x = [&](uint32_t x, uint32_t sw) { S.set(inner_x + inner_of(x, sw), S.get(xin + x)); S.set(sw_x + x / 2, sw); };
        auto conf_y = [&](uint32_t y, uint32_t sw) { S.set(inner_y + inner_of(y, sw), S.get(yin + y)); S.set(sw_y + y / 2, sw); };
        auto take = [&](uint32_t &out) {
            for (uint32_t i = 0; i < nf; i++)
                if (S.get(fr_sw + i)) { out = i; return true; }
            return false;
        };
        // init (sorting.rs:42-64): the single wires
        S.set(inner_y + m - 1, S.get(yin + m - 1));
        if ((m & 1u) == 0) S.set(inner_y + m / 2 - 1, S.get(yin + m - 2));
        else S.set(inner_x + m - 1, S.get(xin + m - 1));
        uint32_t out_idx = m - 1, sw = 0, start = 0, sub = 1;
        bool has_sw = false, has_start = false;
        while (n_free) {
            if (has_sw && S.get(fr_sw + sw)) { S.set(fr_sw + sw, 0u); n_free--; }
            // route_out_wire(out_idx, sub)
            if (!single_y(out_idx)) conf_y(out_idx, sub ^ (out_idx & 1u));
            const uint32_t in_idx = S.get(o_xv + S.get(yin + out_idx));
            if (!single_x(in_idx)) conf_x(in_idx, sub ^ (in_idx & 1u));
            if (single_x(in_idx)) {
                sub ^= 1u;
                has_start = take(start);
                out_idx = has_start ? 2 * start : 0;
                sw = start;
                has_sw = has_start;
                continue;
            }
            // route_in_wire(sibling(in_idx), !sub)
            const uint32_t nx = in_idx + 1 - 2 * (in_idx & 1u), s2 = sub ^ 1u;
            conf_x(nx, s2 ^ (nx & 1u));
            out_idx = S.get(o_yv + S.get(xin + nx));
            if (!single_y(out_idx)) conf_y(out_idx, s2 ^ (out_idx & 1u));
            sw = out_idx / 2;
            has_sw = true;
            if ((has_start && start == sw) || single_y(out_idx)) {
                has_start = take(start);
                out_idx = has_start ? 2 * start : 0;
                sw = start;
                has_sw = has_start;
            } else out_idx = out_idx + 1 - 2 * (out_idx & 1u);
        }
        for (uint32_t i = 0; i < m / 2; i++) S.set(o_bits + n_out++, S.get(sw_x + i));
        for (uint32_t i = 0; i < nf; i++) S.set(o_bits + n_out++, S.get(sw_y + i));
        // the first sub-network is routed first: push the second one below it
        S.set(o_stack + 3 * sp, inner_x + n1); S.set(o_stack + 3 * sp + 1, inner_y + n1); S.set(o_stack + 3 * sp + 2, m - n1);
        sp++;
        S.set(o_stack + 3 * sp, inner_x); S.set(o_stack + 3 * sp + 1, inner_y); S.set(o_stack + 3 * sp + 2, n1);
        sp++;
    }
    // ---- bits.iter().zip(control): insert_value per control bit
    for (uint32_t i = 0; i < n_bits && i < n_out; i++)
        if (!p.insert(bit_ws[2 * i], S.get(o_bits + i) ? fr_one() : fr_zero(), bit_ws[2 * i + 1])) return op_fail(DE_UNSATISFIED);
    return op_ok();
}

}  // namespace acvm
