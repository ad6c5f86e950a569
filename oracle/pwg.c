/*
 * oracle/pwg.c -- TEST INFRASTRUCTURE ONLY (CPU oracle). Not part of the shipped product path.
 * In-order, one-instance-at-a-time restatement of acvm::pwg. Reference anchors per function; see pwg.h.
 */
#include "pwg.h"
#include "hashes.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

void pwg_fail(oracle_acvm_t *a, uint32_t err, uint32_t aux0, uint32_t aux1, const char *msg) {
    a->res.err = err;
    a->res.aux0 = aux0;
    a->res.aux1 = aux1;
    a->res.message[0] = 0;
    if (msg) snprintf(a->res.message, sizeof a->res.message, "%s", msg);
}

static int known(const oracle_acvm_t *a, uint32_t w) {
    if (a->index) return oracle_btree_contains(a->index, w); /* WitnessMap::get = BTreeMap::get (witness_map.rs:47-49) */
    return w < a->nw && a->assigned[w];
}

/* pwg/mod.rs:338-357 insert_value: insert first, then compare with the displaced value */
int pwg_insert_value(oracle_acvm_t *a, uint32_t w, const fr_t *v) {
    if (w >= a->nw) { /* map can hold any key; grow */
        uint32_t nn = w + 1;
        a->val = (fr_t *)realloc(a->val, nn * sizeof(fr_t));
        a->assigned = (uint8_t *)realloc(a->assigned, nn);
        memset(a->assigned + a->nw, 0, nn - a->nw);
        a->nw = nn;
    }
    if (a->index) oracle_btree_insert(a->index, w); /* BTreeMap::insert (witness_map.rs:51-53) */
    if (a->assigned[w]) {
        fr_t old = a->val[w];
        a->val[w] = *v;
        if (!fr_eq(&old, v)) {
            pwg_fail(a, E_UNSATISFIED, 0, 0, NULL);
            return 1;
        }
        return 0;
    }
    a->val[w] = *v;
    a->assigned[w] = 1;
    return 0;
}

/* total / coeff for the constant divisor of opcode a->ip; with ORACLE_MODE_CACHE_INV the inverse is kept per opcode */
static void div_const(oracle_acvm_t *a, fr_t *out, const fr_t *total, const fr_t *coeff) {
    oracle_inv_cache_t *k = a->inv_cache;
    if (!k || a->ip >= a->c->n_opcodes) { fr_div(out, total, coeff); return; }
    if (!k->have[a->ip] || !fr_eq(&k->coeff[a->ip], coeff)) {
        k->coeff[a->ip] = *coeff;
        fr_inverse(&k->inv[a->ip], coeff);
        k->have[a->ip] = 1;
    }
    fr_mul(out, total, &k->inv[a->ip]);
}

/* pwg/mod.rs:309-317 witness_to_value */
static int witness_to_value(oracle_acvm_t *a, uint32_t w, fr_t *out) {
    if (!known(a, w)) {
        pwg_fail(a, E_MISSING_ASSIGNMENT, w, 0, NULL);
        return 1;
    }
    *out = a->val[w];
    return 0;
}

/* arithmetic.rs:212-239 evaluate: partial evaluation with the known witnesses. dst arrays must hold
 * n_mul and n_mul+n_lin entries. */
static void evaluate(const oracle_acvm_t *a, const expr_t *e, expr_t *dst) {
    dst->n_mul = 0;
    dst->n_lin = 0;
    fr_zero(&dst->qc);
    for (size_t i = 0; i < e->n_mul; i++) {
        const mul_term_t *t = &e->mul[i];
        int kl = known(a, t->l), kr = known(a, t->r);
        fr_t v;
        if (kl && kr) { /* MulTerm::Solved(q_m * w_l * w_r) arithmetic.rs:157,227 */
            fr_mul(&v, &t->c, &a->val[t->l]);
            fr_mul(&v, &v, &a->val[t->r]);
            fr_add(&dst->qc, &dst->qc, &v);
        } else if (!kl && !kr) { /* TooManyUnknowns :222-226 */
            if (!fr_is_zero(&t->c)) dst->mul[dst->n_mul++] = *t;
        } else { /* OneUnknown(q_m * known, unknown) :158-159,217-221 */
            uint32_t kw = kl ? t->l : t->r, uw = kl ? t->r : t->l;
            fr_mul(&v, &t->c, &a->val[kw]);
            if (!fr_is_zero(&v)) {
                dst->lin[dst->n_lin].c = v;
                dst->lin[dst->n_lin].w = uw;
                dst->n_lin++;
            }
        }
    }
    for (size_t i = 0; i < e->n_lin; i++) { /* :230-236 */
        const lin_term_t *t = &e->lin[i];
        if (known(a, t->w)) {
            fr_t v;
            fr_mul(&v, &t->c, &a->val[t->w]);
            fr_add(&dst->qc, &dst->qc, &v);
        } else if (!fr_is_zero(&t->c)) {
            dst->lin[dst->n_lin++] = *t;
        }
    }
    fr_add(&dst->qc, &dst->qc, &e->qc); /* :237 */
}

#define EVAL_STACK 8
typedef struct {
    expr_t e;
    mul_term_t mul_s[EVAL_STACK];
    lin_term_t lin_s[2 * EVAL_STACK];
    int heap;
} eval_buf_t;
static void eval_begin(const oracle_acvm_t *a, const expr_t *e, eval_buf_t *b) {
    if (e->n_mul <= EVAL_STACK && e->n_mul + e->n_lin <= 2 * EVAL_STACK) {
        b->e.mul = b->mul_s;
        b->e.lin = b->lin_s;
        b->heap = 0;
    } else {
        b->e.mul = (mul_term_t *)malloc((e->n_mul + 1) * sizeof(mul_term_t));
        b->e.lin = (lin_term_t *)malloc((e->n_mul + e->n_lin + 1) * sizeof(lin_term_t));
        b->heap = 1;
    }
    evaluate(a, e, &b->e);
}
static void eval_end(eval_buf_t *b) {
    if (b->heap) { free(b->e.mul); free(b->e.lin); }
}

/* pwg/mod.rs:321-332 get_value + :362-372 any_witness_from_expression */
int pwg_get_value(oracle_acvm_t *a, const expr_t *e, fr_t *out) {
    eval_buf_t b;
    eval_begin(a, e, &b);
    int rc = 0;
    if (b.e.n_mul == 0 && b.e.n_lin == 0) *out = b.e.qc;
    else {
        uint32_t w = b.e.n_lin ? b.e.lin[0].w : b.e.mul[0].l;
        pwg_fail(a, E_MISSING_ASSIGNMENT, w, 0, NULL);
        rc = 1;
    }
    eval_end(&b);
    return rc;
}

/* arithmetic.rs:27-127 ArithmeticSolver::solve */
static int solve_arithmetic(oracle_acvm_t *a, const expr_t *expr) {
    eval_buf_t b;
    eval_begin(a, expr, &b);
    const expr_t *op = &b.e;
    int rc = 0;
    /* solve_mul_term :133-144 on the evaluated opcode */
    enum { M_SOLVED, M_ONE_UNKNOWN, M_TOO_MANY } mk;
    fr_t m_val; uint32_t m_w = 0;
    fr_zero(&m_val);
    if (op->n_mul == 0) mk = M_SOLVED;
    else if (op->n_mul == 1) {
        const mul_term_t *t = &op->mul[0];
        int kl = known(a, t->l), kr = known(a, t->r);
        if (!kl && !kr) mk = M_TOO_MANY;
        else if (kl && kr) { mk = M_SOLVED; fr_mul(&m_val, &t->c, &a->val[t->l]); fr_mul(&m_val, &m_val, &a->val[t->r]); }
        else { mk = M_ONE_UNKNOWN; m_w = kl ? t->r : t->l; fr_mul(&m_val, &t->c, &a->val[kl ? t->l : t->r]); }
    } else {
        pwg_fail(a, E_PANIC, 0, 0, "Mul term in the arithmetic opcode must contain either zero or one term");
        eval_end(&b);
        return 1;
    }
    /* solve_fan_in_term :176-209 */
    enum { F_SATISFIED, F_SOLVABLE, F_UNSOLVABLE } fk;
    fr_t f_sum, f_coeff; uint32_t f_w = 0; int unknowns = 0;
    fr_zero(&f_sum); fr_zero(&f_coeff);
    fk = F_SATISFIED;
    for (size_t i = 0; i < op->n_lin; i++) {
        const lin_term_t *t = &op->lin[i];
        if (known(a, t->w)) {
            fr_t v; fr_mul(&v, &t->c, &a->val[t->w]); fr_add(&f_sum, &f_sum, &v);
        } else {
            f_coeff = t->c; f_w = t->w; unknowns++;
        }
        if (unknowns > 1) { fk = F_UNSOLVABLE; break; }
    }
    if (fk != F_UNSOLVABLE && unknowns == 1) fk = F_SOLVABLE;

    fr_t total, assignment;
    if (mk == M_TOO_MANY || fk == F_UNSOLVABLE) { /* :38-42 */
        pwg_fail(a, E_TOO_MANY_UNKNOWNS, 0, 0, NULL);
        rc = 1;
    } else if (mk == M_ONE_UNKNOWN && fk == F_SOLVABLE) { /* :43-67 */
        if (m_w == f_w) {
            fr_t den;
            fr_add(&total, &f_sum, &op->qc);
            fr_add(&den, &m_val, &f_coeff);
            if (fr_is_zero(&den)) {
                if (!fr_is_zero(&total)) { pwg_fail(a, E_UNSATISFIED, 0, 0, NULL); rc = 1; }
            } else {
                fr_neg(&total, &total);
                fr_div(&assignment, &total, &den);
                rc = pwg_insert_value(a, m_w, &assignment);
            }
        } else {
            pwg_fail(a, E_TOO_MANY_UNKNOWNS, 0, 0, NULL);
            rc = 1;
        }
    } else if (mk == M_ONE_UNKNOWN && fk == F_SATISFIED) { /* :68-91 */
        fr_add(&total, &f_sum, &op->qc);
        if (fr_is_zero(&m_val)) {
            if (!fr_is_zero(&total)) { pwg_fail(a, E_UNSATISFIED, 0, 0, NULL); rc = 1; }
        } else {
            fr_div(&assignment, &total, &m_val);
            fr_neg(&assignment, &assignment);
            rc = pwg_insert_value(a, m_w, &assignment);
        }
    } else if (mk == M_SOLVED && fk == F_SATISFIED) { /* :92-102 */
        fr_add(&total, &m_val, &f_sum);
        fr_add(&total, &total, &op->qc);
        if (!fr_is_zero(&total)) { pwg_fail(a, E_UNSATISFIED, 0, 0, NULL); rc = 1; }
    } else { /* (Solved, Solvable) :103-125 */
        fr_add(&total, &m_val, &f_sum);
        fr_add(&total, &total, &op->qc);
        if (fr_is_zero(&f_coeff)) {
            if (!fr_is_zero(&total)) { pwg_fail(a, E_UNSATISFIED, 0, 0, NULL); rc = 1; }
        } else {
            div_const(a, &assignment, &total, &f_coeff); /* a / b = a * inverse(b) (generic_ark.rs:375-381): same value */
            fr_neg(&assignment, &assignment);
            rc = pwg_insert_value(a, f_w, &assignment);
        }
    }
    eval_end(&b);
    return rc;
}

/* ------------------------------------------------------------------ black box functions */
static const char *BB_NAMES[BB_COUNT] = {"and", "xor", "range", "sha256", "blake2s", "schnorr_verify", "pedersen",
                                         "hash_to_field_128_security", "ecdsa_secp256k1", "ecdsa_secp256r1",
                                         "fixed_base_scalar_mul", "keccak256", "keccak256", "recursive_aggregation"};
/* BlackBoxFunc reported in errors: Keccak256VariableLength maps to Keccak256 (black_box_function_call.rs:195-197) */
static uint32_t bb_func_of(uint32_t tag) { return tag == BB_KECCAK256_VAR ? BB_KECCAK256 : tag; }

/* blackbox/hash.rs:51-86 get_hash_input */
static int get_hash_input(oracle_acvm_t *a, const bb_call_t *b, int var_len, uint8_t **out, size_t *out_len) {
    size_t cap = 32 * b->n_in[0] + 1, len = 0;
    uint8_t *m = (uint8_t *)malloc(cap);
    for (size_t i = 0; i < b->n_in[0]; i++) {
        fr_t v;
        if (witness_to_value(a, b->in[0][i].witness, &v)) { free(m); return 1; }
        int n = fr_fetch_nearest_bytes(&v, b->in[0][i].num_bits, m + len);
        if (n < 0) { free(m); pwg_fail(a, E_PANIC, 0, 0, "fetch_nearest_bytes: range end index out of range"); return 1; }
        len += (size_t)n;
    }
    if (var_len) {
        fr_t v;
        if (witness_to_value(a, b->in[1][0].witness, &v)) { free(m); return 1; }
        uint64_t lo, hi;
        fr_to_u128(&v, &lo, &hi); /* `to_u128() as usize` truncates to 64 bits */
        if (lo > len) {
            char msg[200];
            snprintf(msg, sizeof msg,
                     "the number of bytes to take from the message is more than the number of bytes in the message. %llu > %zu",
                     (unsigned long long)lo, len);
            free(m);
            pwg_fail(a, E_BLACKBOX_FAILED, BB_KECCAK256, 0, msg);
            return 1;
        }
        len = (size_t)lo;
    }
    *out = m;
    *out_len = len;
    return 0;
}

/* blackbox/hash.rs:28-48,89-103 */
static int solve_hash256(oracle_acvm_t *a, const bb_call_t *b) {
    uint8_t *m; size_t len; uint8_t d[32];
    if (get_hash_input(a, b, b->func == BB_KECCAK256_VAR, &m, &len)) return 1;
    if (b->func == BB_SHA256) oracle_sha256(m, len, d);
    else if (b->func == BB_BLAKE2S) oracle_blake2s(m, len, d);
    else oracle_keccak256(m, len, d);
    free(m);
    if (b->n_out != 32) {
        char msg[64];
        snprintf(msg, sizeof msg, "Expected 32 outputs but encountered %zu", b->n_out);
        pwg_fail(a, E_BLACKBOX_FAILED, bb_func_of(b->func), 0, msg);
        return 1;
    }
    for (int i = 0; i < 32; i++) {
        fr_t v;
        fr_from_be_bytes_reduce(&v, &d[i], 1);
        if (pwg_insert_value(a, b->out[i], &v)) return 1;
    }
    return 0;
}

/* blackbox/signature/mod.rs:5-18 to_u8_vec: last big-endian byte of each witness */
static int to_u8_vec(oracle_acvm_t *a, const func_input_t *in, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; i++) {
        fr_t v; uint8_t be[32];
        if (witness_to_value(a, in[i].witness, &v)) return 1;
        fr_to_be_bytes(&v, be);
        out[i] = be[31];
    }
    return 0;
}

static int backend_rc(oracle_acvm_t *a, int rc, uint32_t func, const char *err) {
    if (rc == 0) return 0;
    if (rc == 1) pwg_fail(a, E_BLACKBOX_FAILED, func, 0, err);       /* pwg/mod.rs:116-127 */
    else if (rc == 2) pwg_fail(a, E_UNSUPPORTED_BLACKBOX, func, 0, NULL);
    else pwg_fail(a, E_PANIC, func, 0, err);
    return 1;
}

/* blackbox/mod.rs:50-163 */
static int solve_blackbox(oracle_acvm_t *a, const bb_call_t *b) {
    /* all-inputs-assigned pre-check (:55-62); get_inputs_vec order (black_box_function_call.rs:205-292) */
    for (int g = 0; g < 4; g++)
        for (size_t i = 0; i < b->n_in[g]; i++)
            if (!known(a, b->in[g][i].witness)) {
                pwg_fail(a, E_MISSING_ASSIGNMENT, b->in[g][i].witness, 0, NULL);
                return 1;
            }
    fr_t x, y, r;
    char err[200];
    err[0] = 0;
    switch (b->func) {
    case BB_AND: case BB_XOR: /* blackbox/logic.rs:11-56 */
        if (b->in[0][0].num_bits != b->in[1][0].num_bits) {
            pwg_fail(a, E_PANIC, 0, 0, "number of bits specified for each input must be the same");
            return 1;
        }
        if (witness_to_value(a, b->in[0][0].witness, &x) || witness_to_value(a, b->in[1][0].witness, &y)) return 1;
        fr_and_xor(&r, &x, &y, b->in[0][0].num_bits, b->func == BB_XOR);
        return pwg_insert_value(a, b->out[0], &r);
    case BB_RANGE: /* blackbox/range.rs:7-18 */
        if (witness_to_value(a, b->in[0][0].witness, &x)) return 1;
        if (fr_num_bits(&x) > b->in[0][0].num_bits) { pwg_fail(a, E_UNSATISFIED, 0, 0, NULL); return 1; }
        return 0;
    case BB_SHA256: case BB_BLAKE2S: case BB_KECCAK256: case BB_KECCAK256_VAR:
        return solve_hash256(a, b);
    case BB_HASH_TO_FIELD_128: { /* blackbox/hash.rs:13-24; blackbox_solver/src/lib.rs:62-65,94-99 */
        uint8_t *m; size_t len; uint8_t d[32];
        if (get_hash_input(a, b, 0, &m, &len)) return 1;
        oracle_blake2s(m, len, d);
        free(m);
        fr_from_be_bytes_reduce(&r, d, 32);
        return pwg_insert_value(a, b->out[0], &r);
    }
    case BB_SCHNORR_VERIFY: { /* blackbox/signature/schnorr.rs:13-35 */
        if (witness_to_value(a, b->in[0][0].witness, &x) || witness_to_value(a, b->in[1][0].witness, &y)) return 1;
        uint8_t *sig = (uint8_t *)malloc(b->n_in[2] + 1), *msg = (uint8_t *)malloc(b->n_in[3] + 1);
        int rc = to_u8_vec(a, b->in[2], b->n_in[2], sig) || to_u8_vec(a, b->in[3], b->n_in[3], msg);
        int ok = 0;
        if (!rc) rc = backend_rc(a, a->backend->schnorr_verify(a->backend->ctx, &x, &y, sig, b->n_in[2], msg, b->n_in[3], &ok, err, sizeof err),
                                 BB_SCHNORR_VERIFY, err);
        free(sig); free(msg);
        if (rc) return 1;
        fr_from_u64(&r, ok ? 1 : 0);
        return pwg_insert_value(a, b->out[0], &r);
    }
    case BB_PEDERSEN: { /* blackbox/pedersen.rs:11-28 */
        fr_t *sc = (fr_t *)malloc((b->n_in[0] + 1) * sizeof(fr_t));
        for (size_t i = 0; i < b->n_in[0]; i++)
            if (witness_to_value(a, b->in[0][i].witness, &sc[i])) { free(sc); return 1; }
        int rc = backend_rc(a, a->backend->pedersen(a->backend->ctx, sc, b->n_in[0], b->domain_separator, &x, &y, err, sizeof err),
                            BB_PEDERSEN, err);
        free(sc);
        if (rc) return 1;
        if (pwg_insert_value(a, b->out[0], &x)) return 1;
        return pwg_insert_value(a, b->out[1], &y);
    }
    case BB_FIXED_BASE_SCALAR_MUL: { /* blackbox/fixed_base_scalar_mul.rs:11-27 */
        fr_t lo, hi;
        if (witness_to_value(a, b->in[0][0].witness, &lo) || witness_to_value(a, b->in[1][0].witness, &hi)) return 1;
        if (backend_rc(a, a->backend->fixed_base_scalar_mul(a->backend->ctx, &lo, &hi, &x, &y, err, sizeof err),
                       BB_FIXED_BASE_SCALAR_MUL, err)) return 1;
        if (pwg_insert_value(a, b->out[0], &x)) return 1;
        return pwg_insert_value(a, b->out[1], &y);
    }
    case BB_ECDSA_SECP256K1: case BB_ECDSA_SECP256R1: { /* blackbox/signature/ecdsa.rs:12-91 */
        /* hashed_message first, then pubkey_x / pubkey_y / signature with their length checks (ecdsa.rs:20-47) */
        size_t nm = b->n_in[3];
        uint8_t *msg = (uint8_t *)malloc(nm + 1), px[33], py[33], sg[65];
        int rc = to_u8_vec(a, b->in[3], nm, msg);
        static const char *what[3] = {"pubkey_x", "pubkey_y", "signature"};
        static const size_t want[3] = {32, 32, 64};
        uint8_t *dst[3] = {px, py, sg};
        for (int g = 0; g < 3 && !rc; g++) {
            if (b->n_in[g] != want[g]) {
                char m[96];
                snprintf(m, sizeof m, "expected %s size %zu but received %zu", what[g], want[g], b->n_in[g]);
                pwg_fail(a, E_BLACKBOX_FAILED, b->func, 0, m);
                rc = 1;
            } else rc = to_u8_vec(a, b->in[g], b->n_in[g], dst[g]);
        }
        if (!rc) {
            int v = oracle_ecdsa_verify(b->func == BB_ECDSA_SECP256R1, msg, nm, px, py, sg);
            if (v < 0) { pwg_fail(a, E_PANIC, b->func, (uint32_t)(-v), oracle_ecdsa_panic_text(v)); rc = 1; }
            else { fr_from_u64(&r, (uint64_t)v); rc = pwg_insert_value(a, b->out[0], &r); }
        }
        free(msg);
        return rc;
    }
    case BB_RECURSIVE_AGGREGATION: /* blackbox/mod.rs:154-161 */
        fr_zero(&r);
        for (size_t i = 0; i < b->n_out; i++)
            if (pwg_insert_value(a, b->out[i], &r)) return 1;
        return 0;
    }
    (void)BB_NAMES;
    pwg_fail(a, E_PANIC, 0, 0, "unknown black box function");
    return 1;
}

/* ------------------------------------------------------------------ directives */
/* 256-bit unsigned long division, canonical integers (num-bigint semantics, directives/mod.rs:28-59) */
static void divrem256(const uint64_t a[4], const uint64_t b[4], uint64_t q[4], uint64_t r[4]) {
    memset(q, 0, 32);
    memset(r, 0, 32);
    for (int i = 255; i >= 0; i--) {
        /* r = (r << 1) | bit_i(a) */
        r[3] = (r[3] << 1) | (r[2] >> 63);
        r[2] = (r[2] << 1) | (r[1] >> 63);
        r[1] = (r[1] << 1) | (r[0] >> 63);
        r[0] = (r[0] << 1) | ((a[i / 64] >> (i % 64)) & 1);
        int ge = 1;
        for (int k = 3; k >= 0; k--) {
            if (r[k] > b[k]) break;
            if (r[k] < b[k]) { ge = 0; break; }
        }
        if (ge) {
            uint64_t borrow = 0;
            for (int k = 0; k < 4; k++) {
                unsigned __int128 d = (unsigned __int128)r[k] - b[k] - borrow;
                r[k] = (uint64_t)d;
                borrow = (uint64_t)(d >> 64) & 1;
            }
            q[i / 64] |= 1ULL << (i % 64);
        }
    }
}

static int solve_directive(oracle_acvm_t *a, const directive_t *d) {
    if (d->kind == DIR_QUOTIENT) {
        fr_t va, vb, pred;
        if (pwg_get_value(a, &d->a, &va) || pwg_get_value(a, &d->b, &vb)) return 1;
        if (d->has_predicate) { if (pwg_get_value(a, &d->predicate, &pred)) return 1; }
        else fr_one(&pred);
        uint64_t ia[4], ib[4], q[4] = {0, 0, 0, 0}, r[4] = {0, 0, 0, 0};
        fr_to_canonical(&va, ia);
        fr_to_canonical(&vb, ib);
        if (!(fr_is_zero(&pred) || fr_is_zero(&vb))) divrem256(ia, ib, q, r);
        fr_t fq, frr;
        fr_from_canonical(&fq, q); /* q,r <= a < p */
        fr_from_canonical(&frr, r);
        if (pwg_insert_value(a, d->q, &fq)) return 1;
        return pwg_insert_value(a, d->r, &frr);
    }
    if (d->kind == DIR_TO_LE_RADIX) { /* directives/mod.rs:60-87 */
        fr_t va;
        if (pwg_get_value(a, &d->a, &va)) return 1;
        if (d->radix < 2 || d->radix > 256) {
            pwg_fail(a, E_PANIC, 0, 0, "The radix must be within 2...256"); /* num-bigint to_radix_le assert */
            return 1;
        }
        uint64_t v[4];
        fr_to_canonical(&va, v);
        uint8_t digits[256];
        size_t nd = 0;
        if (!(v[0] | v[1] | v[2] | v[3])) digits[nd++] = 0; /* 0 -> [0] */
        while (v[0] | v[1] | v[2] | v[3]) {
            unsigned __int128 rem = 0;
            for (int k = 3; k >= 0; k--) {
                unsigned __int128 cur = (rem << 64) | v[k];
                v[k] = (uint64_t)(cur / d->radix);
                rem = cur % d->radix;
            }
            digits[nd++] = (uint8_t)rem;
        }
        if (d->n_bw < nd) { pwg_fail(a, E_UNSATISFIED, 0, 0, NULL); return 1; }
        for (size_t i = 0; i < d->n_bw; i++) {
            fr_t dv;
            if (i < nd) fr_from_be_bytes_reduce(&dv, &digits[i], 1);
            else fr_zero(&dv);
            if (pwg_insert_value(a, d->bw[i], &dv)) return 1;
        }
        return 0;
    }
    return oracle_solve_permutation_sort(a, d); /* sorting.c */
}

/* ------------------------------------------------------------------ memory (memory_op.rs) */
static mem_block_t *block_entry(oracle_acvm_t *a, uint32_t id) { /* block_solvers.entry(id).or_default() */
    for (size_t i = 0; i < a->n_blocks; i++)
        if (a->blocks[i].id == id) return &a->blocks[i];
    a->blocks = (mem_block_t *)realloc(a->blocks, (a->n_blocks + 1) * sizeof(mem_block_t));
    mem_block_t *b = &a->blocks[a->n_blocks++];
    memset(b, 0, sizeof *b);
    b->id = id;
    return b;
}
static int mem_write(oracle_acvm_t *a, mem_block_t *b, uint32_t index, const fr_t *v) { /* :21-35 */
    if (index >= b->len) { pwg_fail(a, E_INDEX_OOB, index, b->len, NULL); return 1; }
    if (index >= b->cap) {
        uint32_t nc = b->len > index + 1 ? b->len : index + 1;
        b->cells = (fr_t *)realloc(b->cells, nc * sizeof(fr_t));
        b->present = (uint8_t *)realloc(b->present, nc);
        memset(b->present + b->cap, 0, nc - b->cap);
        b->cap = nc;
    }
    b->cells[index] = *v;
    b->present[index] = 1;
    return 0;
}
static int mem_read(oracle_acvm_t *a, mem_block_t *b, uint32_t index, fr_t *out) { /* :37-44: key presence only */
    if (index < b->cap && b->present[index]) { *out = b->cells[index]; return 0; }
    pwg_fail(a, E_INDEX_OOB, index, b->len, NULL);
    return 1;
}
static int solve_memory_init(oracle_acvm_t *a, const opcode_t *o) { /* :47-60 */
    mem_block_t *b = block_entry(a, o->block_id);
    b->len = (uint32_t)o->n_init;
    for (size_t i = 0; i < o->n_init; i++) {
        fr_t v;
        if (witness_to_value(a, o->init[i], &v)) return 1;
        if (mem_write(a, b, (uint32_t)i, &v)) return 1;
    }
    return 0;
}
static int solve_memory_op(oracle_acvm_t *a, const opcode_t *o) { /* :62-124 */
    mem_block_t *b = block_entry(a, o->block_id);
    fr_t operation, index, pred;
    if (pwg_get_value(a, &o->mem_operation, &operation)) return 1;
    if (pwg_get_value(a, &o->mem_index, &index)) return 1;
    uint64_t idx64;
    if (!fr_try_to_u64(&index, &idx64)) { /* :72 try_to_u64().unwrap() */
        pwg_fail(a, E_PANIC, 0, 0, "called `Option::unwrap()` on a `None` value (memory index)");
        return 1;
    }
    uint32_t mi = (uint32_t)idx64; /* `as MemoryIndex` wraps */
    eval_buf_t vb;
    eval_begin(a, &o->mem_value, &vb);
    int is_read = fr_is_zero(&operation);
    int rc = 0;
    if (o->has_predicate) { if (pwg_get_value(a, &o->predicate, &pred)) { eval_end(&vb); return 1; } }
    else fr_one(&pred);
    if (is_read) {
        /* Expression::to_witness (expression/mod.rs:158-172): one linear term, coeff 1, constant 0, no mul */
        const expr_t *v = &vb.e;
        if (!(v->n_mul == 0 && v->n_lin == 1 && fr_is_one(&v->lin[0].c) && fr_is_zero(&v->qc))) {
            pwg_fail(a, E_PANIC, 0, 0, "Memory must be read into a specified witness index, encountered an Expression");
            rc = 1;
        } else {
            uint32_t w = v->lin[0].w;
            fr_t val;
            if (fr_is_zero(&pred)) fr_zero(&val);
            else rc = mem_read(a, b, mi, &val);
            if (!rc) rc = pwg_insert_value(a, w, &val);
        }
    } else if (!fr_is_zero(&pred)) {
        /* get_value(&value_write): re-evaluates the already-evaluated expression -> same result */
        const expr_t *v = &vb.e;
        if (v->n_mul == 0 && v->n_lin == 0) rc = mem_write(a, b, mi, &v->qc);
        else {
            pwg_fail(a, E_MISSING_ASSIGNMENT, v->n_lin ? v->lin[0].w : v->mul[0].l, 0, NULL);
            rc = 1;
        }
    }
    eval_end(&vb);
    return rc;
}

/* ------------------------------------------------------------------ ACVM */
oracle_inv_cache_t *oracle_inv_cache_new(const circuit_t *c) {
    oracle_inv_cache_t *k = (oracle_inv_cache_t *)calloc(1, sizeof *k);
    k->coeff = (fr_t *)calloc(c->n_opcodes + 1, sizeof(fr_t));
    k->inv = (fr_t *)calloc(c->n_opcodes + 1, sizeof(fr_t));
    k->have = (uint8_t *)calloc(c->n_opcodes + 1, 1);
    return k;
}
void oracle_inv_cache_free(oracle_inv_cache_t *k) {
    if (!k) return;
    free(k->coeff); free(k->inv); free(k->have); free(k);
}
oracle_acvm_t *oracle_acvm_new(const circuit_t *c, const backend_t *backend, size_t n_initial,
                               const uint32_t *ids, const uint8_t *values_be32) {
    return oracle_acvm_new_mode(c, backend, n_initial, ids, values_be32, 0, NULL);
}
oracle_acvm_t *oracle_acvm_new_mode(const circuit_t *c, const backend_t *backend, size_t n_initial, const uint32_t *ids,
                                    const uint8_t *values_be32, int mode, oracle_inv_cache_t *cache) {
    oracle_acvm_t *a = (oracle_acvm_t *)calloc(1, sizeof *a);
    a->mode = mode;
    a->index = (mode & ORACLE_MODE_SPARSE_MAP) ? oracle_btree_new() : NULL;
    a->inv_cache = (mode & ORACLE_MODE_CACHE_INV) ? cache : NULL;
    a->c = c;
    a->backend = backend ? backend : oracle_backend(0);
    uint32_t nw = c->max_witness + 1;
    for (size_t i = 0; i < n_initial; i++)
        if (ids[i] + 1 > nw) nw = ids[i] + 1;
    a->nw = nw;
    a->val = (fr_t *)calloc(nw, sizeof(fr_t));
    a->assigned = (uint8_t *)calloc(nw, 1);
    for (size_t i = 0; i < n_initial; i++) {
        fr_from_be_bytes_reduce(&a->val[ids[i]], values_be32 + 32 * i, 32);
        a->assigned[ids[i]] = 1;
        if (a->index) oracle_btree_insert(a->index, ids[i]);
    }
    a->res.status = c->n_opcodes == 0 ? ST_SOLVED : ST_IN_PROGRESS; /* pwg/mod.rs:147 */
    a->extra_fc = (fc_result_t **)calloc(c->n_opcodes + 1, sizeof(fc_result_t *));
    a->n_extra_fc = (size_t *)calloc(c->n_opcodes + 1, sizeof(size_t));
    return a;
}

static void free_pending(oracle_acvm_t *a) {
    free(a->pending.function);
    for (size_t i = 0; i < a->pending.n_inputs; i++) free(a->pending.inputs[i]);
    free(a->pending.inputs);
    free(a->pending.input_len);
    memset(&a->pending, 0, sizeof a->pending);
}

/* The state of ACVM::new (pwg/mod.rs:146-156) for the NEXT instance of the same circuit on the same object: what oracle_acvm_free +
 * oracle_acvm_new_mode would leave, without handing the witness vector (32 B x witnesses: 320 KB for a 10k-gate circuit, an mmap of its
 * own) back to the OS and faulting it in again per instance -- under the process-wide mm lock, which is what held the multi-threaded CPU
 * baseline to 14 x one thread on 64 cores. The reference's caller owns one ACVM per execution; a batch driver on the CPU would reuse its
 * buffers exactly like this. ids must be the ones the object was created with (or any below its witness count). */
void oracle_acvm_reset(oracle_acvm_t *a, size_t n_initial, const uint32_t *ids, const uint8_t *values_be32) {
    for (size_t i = 0; i < a->n_blocks; i++) { free(a->blocks[i].cells); free(a->blocks[i].present); }
    free(a->blocks);
    a->blocks = NULL;
    a->n_blocks = 0;
    for (size_t i = 0; i < a->c->n_opcodes; i++) {
        if (!a->n_extra_fc[i]) continue;
        for (size_t j = 0; j < a->n_extra_fc[i]; j++) {
            for (size_t k = 0; k < a->extra_fc[i][j].n; k++) free(a->extra_fc[i][j].values[k].arr);
            free(a->extra_fc[i][j].values);
        }
        free(a->extra_fc[i]);
        a->extra_fc[i] = NULL;
        a->n_extra_fc[i] = 0;
    }
    free_pending(a);
    memset(a->val, 0, (size_t)a->nw * sizeof(fr_t));
    memset(a->assigned, 0, a->nw);
    if (a->index) { /* the sparse-map mode times a map that starts empty, like the reference's */
        oracle_btree_free(a->index);
        a->index = oracle_btree_new();
    }
    for (size_t i = 0; i < n_initial; i++) {
        if (ids[i] >= a->nw) continue;
        fr_from_be_bytes_reduce(&a->val[ids[i]], values_be32 + 32 * i, 32);
        a->assigned[ids[i]] = 1;
        if (a->index) oracle_btree_insert(a->index, ids[i]);
    }
    a->ip = 0;
    memset(&a->res, 0, sizeof a->res);
    a->res.status = a->c->n_opcodes == 0 ? ST_SOLVED : ST_IN_PROGRESS;
}

void oracle_acvm_free(oracle_acvm_t *a) {
    if (!a) return;
    for (size_t i = 0; i < a->n_blocks; i++) { free(a->blocks[i].cells); free(a->blocks[i].present); }
    free(a->blocks);
    for (size_t i = 0; i < a->c->n_opcodes; i++) {
        for (size_t j = 0; j < a->n_extra_fc[i]; j++) {
            for (size_t k = 0; k < a->extra_fc[i][j].n; k++) free(a->extra_fc[i][j].values[k].arr);
            free(a->extra_fc[i][j].values);
        }
        free(a->extra_fc[i]);
    }
    free(a->extra_fc); free(a->n_extra_fc);
    free_pending(a);
    free(a->val); free(a->assigned);
    oracle_btree_free(a->index);
    free(a);
}

uint32_t oracle_acvm_solve_opcode(oracle_acvm_t *a) {
    const opcode_t *o = &a->c->opcodes[a->ip];
    int rc;
    a->res.err = E_NONE;
    switch (o->kind) {
    case OP_ARITHMETIC: rc = solve_arithmetic(a, &o->expr); break;
    case OP_BLACKBOX: rc = solve_blackbox(a, &o->bb); break;
    case OP_DIRECTIVE: rc = solve_directive(a, &o->dir); break;
    case OP_MEMORY_INIT: rc = solve_memory_init(a, o); break;
    case OP_MEMORY_OP: rc = solve_memory_op(a, o); break;
    case OP_BRILLIG:
        rc = brillig_solve(a, &o->brillig, a->ip);
        if (rc == 2) { /* pwg/mod.rs:267: ip NOT advanced */
            a->res.status = ST_REQUIRES_FOREIGN_CALL;
            return a->res.status;
        }
        break;
    default: pwg_fail(a, E_PANIC, 0, 0, "bad opcode"); rc = 1;
    }
    if (!rc) { /* :273-279 */
        a->ip++;
        a->res.status = a->ip == a->c->n_opcodes ? ST_SOLVED : ST_IN_PROGRESS;
    } else { /* :281-301 */
        a->res.opcode_index = (uint32_t)a->ip;
        a->res.status = ST_FAILURE;
    }
    return a->res.status;
}

uint32_t oracle_acvm_solve(oracle_acvm_t *a) {
    while (a->res.status == ST_IN_PROGRESS) oracle_acvm_solve_opcode(a);
    return a->res.status;
}

int oracle_acvm_resolve_foreign_call(oracle_acvm_t *a, const fc_result_t *result) {
    if (a->res.status != ST_REQUIRES_FOREIGN_CALL) return -1; /* reference panics (:215-217) */
    size_t i = a->ip, n = a->n_extra_fc[i];
    a->extra_fc[i] = (fc_result_t *)realloc(a->extra_fc[i], (n + 1) * sizeof(fc_result_t));
    fc_result_t *dst = &a->extra_fc[i][n];
    dst->n = result->n;
    dst->values = (fc_output_t *)calloc(result->n ? result->n : 1, sizeof(fc_output_t));
    for (size_t k = 0; k < result->n; k++) {
        dst->values[k] = result->values[k];
        if (result->values[k].is_array) {
            dst->values[k].arr = (fr_t *)malloc((result->values[k].n + 1) * sizeof(fr_t));
            memcpy(dst->values[k].arr, result->values[k].arr, result->values[k].n * sizeof(fr_t));
        }
    }
    a->n_extra_fc[i] = n + 1;
    free_pending(a);
    a->res.status = ST_IN_PROGRESS;
    return 0;
}
