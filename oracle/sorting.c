/*
 * oracle/sorting.c -- TEST INFRASTRUCTURE ONLY (CPU oracle). Not part of the shipped product path.
 *
 * Directive::PermutationSort:
 *   /root/reference/acvm/src/pwg/directives/mod.rs:88-119      (tuples, stable sort by `sort_by`, control bits -> witnesses)
 *   /root/reference/acvm/src/pwg/directives/sorting.rs:8-244   (SortingNetwork + route: control bits of the permutation
 *                                                               network that maps `inputs` to `outputs`)
 * The reference keys its maps by FieldElement; every caller passes distinct values (the element indices), so values are
 * restated as distinct uint32 ids and the BTreeMaps as arrays indexed by value; `free` (a BTreeSet whose smallest element is
 * taken) is a bitmap scanned from 0. PINNING: the five known answers of sorting.rs:309-383 and its network-execution property
 * (tests/test_oracle_sorting.py).
 */
#include "pwg.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    uint32_t n;
    const uint32_t *x_inputs, *y_inputs;
    int32_t *x_values, *y_values; /* value -> index, -1 once removed */
    uint32_t *inner_x, *inner_y;
    uint8_t *switch_x, *switch_y, *free_sw;
    uint32_t n_free;
} network_t;

static int is_single_x(const network_t *w, uint32_t a) { return w->n % 2 == 1 && a == w->n - 1; }
static int is_single_y(const network_t *w, uint32_t a) { return a >= w->n - 2 + w->n % 2; }
static uint32_t compute_inner(const network_t *w, uint32_t idx, int sw) { return (sw ^ (idx % 2 == 1)) ? idx / 2 + w->n / 2 : idx / 2; }
static void configure_x(network_t *w, uint32_t x, int sw, uint32_t inner) { w->inner_x[inner] = w->x_inputs[x]; w->switch_x[x / 2] = (uint8_t)sw; }
static void configure_y(network_t *w, uint32_t y, int sw, uint32_t inner) { w->inner_y[inner] = w->y_inputs[y]; w->switch_y[y / 2] = (uint8_t)sw; }
static uint32_t sibling(uint32_t i) { return i + 1 - 2 * (i % 2); }

static uint32_t route_out_wire(network_t *w, uint32_t y, int sub) { /* sorting.rs:67-86 */
    if (!is_single_y(w, y)) {
        int s1 = sub ^ (y % 2 != 0);
        configure_y(w, y, s1, compute_inner(w, y, s1));
    }
    uint32_t v = w->y_inputs[y];
    uint32_t x = (uint32_t)w->x_values[v];
    w->x_values[v] = -1;
    if (!is_single_x(w, x)) {
        int s2 = sub ^ (x % 2 != 0);
        configure_x(w, x, s2, compute_inner(w, x, s2));
    }
    return x;
}
static uint32_t route_in_wire(network_t *w, uint32_t x, int sub) { /* sorting.rs:89-107 */
    int s1 = sub ^ (x % 2 != 0);
    configure_x(w, x, s1, compute_inner(w, x, s1));
    uint32_t v = w->x_inputs[x];
    uint32_t y = (uint32_t)w->y_values[v];
    w->y_values[v] = -1;
    if (!is_single_y(w, y)) {
        int s2 = sub ^ (y % 2 != 0);
        configure_y(w, y, s2, compute_inner(w, y, s2));
    }
    return y;
}
static int take(const network_t *w, uint32_t *out) { /* free.first() */
    uint32_t nf = (w->n - 1) / 2;
    for (uint32_t i = 0; i < nf; i++)
        if (w->free_sw[i]) { *out = i; return 1; }
    return 0;
}

/* sorting.rs:161-244. values: distinct ids < max_value. bits: appended at *n_bits. */
static void route(const uint32_t *inputs, const uint32_t *outputs, uint32_t n, uint32_t max_value, uint8_t *bits, size_t *n_bits) {
    if (n <= 1) return;
    if (n == 2) { bits[(*n_bits)++] = inputs[0] != outputs[0]; return; }
    uint32_t n1 = n / 2, nf = (n - 1) / 2;
    network_t w;
    w.n = n; w.x_inputs = inputs; w.y_inputs = outputs;
    w.x_values = (int32_t *)malloc(max_value * sizeof(int32_t));
    w.y_values = (int32_t *)malloc(max_value * sizeof(int32_t));
    w.inner_x = (uint32_t *)calloc(n, sizeof(uint32_t));
    w.inner_y = (uint32_t *)calloc(n, sizeof(uint32_t));
    w.switch_x = (uint8_t *)calloc(n / 2 + 1, 1);
    w.switch_y = (uint8_t *)calloc(nf + 1, 1);
    w.free_sw = (uint8_t *)malloc(nf + 1);
    memset(w.free_sw, 1, nf);
    w.n_free = nf;
    for (uint32_t i = 0; i < n; i++) { w.x_values[inputs[i]] = (int32_t)i; w.y_values[outputs[i]] = (int32_t)i; }
    /* init :42-64: route the single wires */
    w.inner_y[n - 1] = outputs[n - 1];
    if (n % 2 == 0) w.inner_y[n / 2 - 1] = outputs[n - 2];
    else w.inner_x[n - 1] = inputs[n - 1];
    uint32_t out_idx = n - 1, sw = 0, start = 0;
    int start_sub = 1, has_sw = 0, has_start = 0;
    while (w.n_free) {
        if (has_sw && w.free_sw[sw]) { w.free_sw[sw] = 0; w.n_free--; }
        uint32_t in_idx = route_out_wire(&w, out_idx, start_sub);
        if (is_single_x(&w, in_idx)) {
            start_sub = !start_sub;
            has_start = take(&w, &start);
            out_idx = has_start ? 2 * start : 0;
            sw = start; has_sw = has_start;
            continue;
        }
        out_idx = route_in_wire(&w, sibling(in_idx), !start_sub);
        sw = out_idx / 2; has_sw = 1;
        if ((has_start && start == sw) || is_single_y(&w, out_idx)) {
            has_start = take(&w, &start);
            out_idx = has_start ? 2 * start : 0;
            sw = start; has_sw = has_start;
        } else out_idx = sibling(out_idx);
    }
    for (uint32_t i = 0; i < n / 2; i++) bits[(*n_bits)++] = w.switch_x[i];
    for (uint32_t i = 0; i < nf; i++) bits[(*n_bits)++] = w.switch_y[i];
    route(w.inner_x, w.inner_y, n1, max_value, bits, n_bits);
    route(w.inner_x + n1, w.inner_y + n1, n - n1, max_value, bits, n_bits);
    free(w.x_values); free(w.y_values); free(w.inner_x); free(w.inner_y); free(w.switch_x); free(w.switch_y); free(w.free_sw);
}

/* exported for the tests: control bits of the network mapping inputs -> outputs (distinct values < 65536). Returns the count. */
size_t oracle_sorting_route(const uint32_t *inputs, const uint32_t *outputs, uint32_t n, uint8_t *bits) {
    size_t nb = 0;
    uint32_t mx = 0;
    for (uint32_t i = 0; i < n; i++) { if (inputs[i] + 1 > mx) mx = inputs[i] + 1; if (outputs[i] + 1 > mx) mx = outputs[i] + 1; }
    route(inputs, outputs, n, mx + 1, bits, &nb);
    return nb;
}

/* directives/mod.rs:88-119 */
int oracle_solve_permutation_sort(oracle_acvm_t *a, const directive_t *d) {
    size_t n = d->n_sort_inputs, tuple = d->tuple;
    fr_t *vals = (fr_t *)malloc((n * tuple + 1) * sizeof(fr_t));
    uint32_t *order = (uint32_t *)malloc((n + 1) * sizeof(uint32_t)), *base = (uint32_t *)malloc((n + 1) * sizeof(uint32_t));
    int rc = 0;
    for (size_t i = 0; i < n && !rc; i++) {
        if (d->sort_input_len[i] != tuple) { pwg_fail(a, E_PANIC, 0, 0, "assertion failed: element.len() == tuple"); rc = 1; break; }
        for (size_t k = 0; k < tuple; k++)
            if (pwg_get_value(a, &d->sort_inputs[i][k], &vals[i * tuple + k])) { rc = 1; break; }
        order[i] = (uint32_t)i;
        base[i] = (uint32_t)i;
    }
    if (!rc) {
        /* stable sort by the columns of sort_by, compared as integers (sort_by :102-112); insertion sort is stable */
        for (size_t i = 1; i < n; i++) {
            uint32_t cur = order[i];
            size_t j = i;
            while (j > 0) {
                int cmp = 0;
                for (size_t s = 0; s < d->n_sort_by && !cmp; s++) {
                    uint32_t col = d->sort_by[s];
                    /* a[*i as usize] on a Vec of tuple + 1 elements (mod.rs:102-105) panics once the comparator reaches a column
                     * beyond it: with the bad column FIRST in sort_by that is the first comparison of any sort of n >= 2 elements
                     * (the case the tests pin); behind other columns it needs a compared pair that ties on all of them, and which
                     * pairs Rust's sort compares is its implementation's business -- here it is this insertion sort's pairs. */
                    if (col > tuple) {
                        char msg[96];
                        snprintf(msg, sizeof msg, "index out of bounds: the len is %u but the index is %u", (unsigned)(tuple + 1), (unsigned)col);
                        pwg_fail(a, E_PANIC, 0, 0, msg);
                        rc = 1;
                        break;
                    }
                    uint64_t x[4], y[4];
                    if (col == tuple) /* column `tuple` is the index itself */ { cmp = (order[j - 1] > cur) - (order[j - 1] < cur); continue; }
                    fr_to_canonical(&vals[order[j - 1] * tuple + col], x);
                    fr_to_canonical(&vals[cur * tuple + col], y);
                    for (int k = 3; k >= 0 && !cmp; k--) cmp = (x[k] > y[k]) - (x[k] < y[k]);
                }
                if (rc || cmp <= 0) break;
                order[j] = order[j - 1];
                j--;
            }
            if (rc) break;
            order[j] = cur;
        }
    }
    if (!rc) {
        uint8_t *bits = (uint8_t *)calloc(n * 32 + 8, 1);
        size_t nb = 0;
        route(base, order, (uint32_t)n, (uint32_t)n + 1, bits, &nb);
        for (size_t i = 0; i < d->n_bw && i < nb && !rc; i++) { /* bits.iter().zip(control) */
            fr_t v;
            fr_from_u64(&v, bits[i]);
            rc = pwg_insert_value(a, d->bw[i], &v);
        }
        free(bits);
    }
    free(vals); free(order); free(base);
    return rc;
}
