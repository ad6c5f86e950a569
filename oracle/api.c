/*
 * oracle/api.c -- TEST INFRASTRUCTURE ONLY (CPU oracle). Not part of the shipped product path.
 * ctypes-facing entry points of liboracle.so. Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (libacvm_amd.so) never links or calls it.
 */
#include "hashes.h"
#include "pwg.h"
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

/* Circuit::read (acir/src/circuit/mod.rs:154-161): gzip + bincode. Accepts raw bincode as well. */
circuit_t *oracle_circuit_from_bytes(const uint8_t *buf, size_t len) {
    if (len >= 2 && buf[0] == 0x1f && buf[1] == 0x8b) {
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, 16 + MAX_WBITS) != Z_OK) return NULL;
        size_t cap = len * 8 + 1024, out_len = 0;
        uint8_t *out = (uint8_t *)malloc(cap);
        zs.next_in = (Bytef *)buf;
        zs.avail_in = (uInt)len;
        int rc;
        do {
            if (out_len == cap) { cap *= 2; out = (uint8_t *)realloc(out, cap); }
            zs.next_out = out + out_len;
            zs.avail_out = (uInt)(cap - out_len);
            rc = inflate(&zs, Z_NO_FLUSH);
            out_len = cap - zs.avail_out;
        } while (rc == Z_OK);
        inflateEnd(&zs);
        circuit_t *c = rc == Z_STREAM_END ? acir_circuit_parse(out, out_len) : NULL;
        free(out);
        return c;
    }
    return acir_circuit_parse(buf, len);
}
void oracle_circuit_free(circuit_t *c) { acir_circuit_free(c); }
uint32_t oracle_circuit_num_witnesses(const circuit_t *c) { return c->max_witness + 1; }
uint32_t oracle_circuit_num_opcodes(const circuit_t *c) { return (uint32_t)c->n_opcodes; }
uint32_t oracle_circuit_current_witness_index(const circuit_t *c) { return c->current_witness_index; }
size_t oracle_result_size(void) { return sizeof(acvm_result_t); }

/* ---- single instance handle (mirrors the ACVM call shape, pwg/mod.rs:145-304) ---- */
oracle_acvm_t *oracle_acvm_create(const circuit_t *c, int backend, size_t n_initial, const uint32_t *ids,
                                  const uint8_t *values_be32) {
    return oracle_acvm_new(c, oracle_backend(backend), n_initial, ids, values_be32);
}
void oracle_acvm_destroy(oracle_acvm_t *a) { oracle_acvm_free(a); }
uint32_t oracle_acvm_run(oracle_acvm_t *a) { return oracle_acvm_solve(a); }
uint32_t oracle_acvm_step(oracle_acvm_t *a) { return oracle_acvm_solve_opcode(a); }
void oracle_acvm_result(const oracle_acvm_t *a, acvm_result_t *out) { *out = a->res; }
uint32_t oracle_acvm_instruction_pointer(const oracle_acvm_t *a) { return (uint32_t)a->ip; }
uint32_t oracle_acvm_num_witnesses(const oracle_acvm_t *a) { return a->nw; }
/* witness_map(): assigned[nw] flags + values[nw][32] canonical big-endian (zeros where unassigned) */
void oracle_acvm_witness_map(const oracle_acvm_t *a, uint8_t *assigned, uint8_t *values_be32) {
    for (uint32_t w = 0; w < a->nw; w++) {
        assigned[w] = a->assigned[w];
        if (a->assigned[w]) fr_to_be_bytes(&a->val[w], values_be32 + 32 * (size_t)w);
        else memset(values_be32 + 32 * (size_t)w, 0, 32);
    }
}
/* get_pending_foreign_call (:203-209) */
const char *oracle_acvm_pending_function(const oracle_acvm_t *a) {
    return a->res.status == ST_REQUIRES_FOREIGN_CALL ? a->pending.function : NULL;
}
uint32_t oracle_acvm_pending_num_inputs(const oracle_acvm_t *a) { return (uint32_t)a->pending.n_inputs; }
uint32_t oracle_acvm_pending_input_len(const oracle_acvm_t *a, uint32_t i) { return (uint32_t)a->pending.input_len[i]; }
void oracle_acvm_pending_input(const oracle_acvm_t *a, uint32_t i, uint8_t *out_be32) {
    for (size_t k = 0; k < a->pending.input_len[i]; k++) fr_to_be_bytes(&a->pending.inputs[i][k], out_be32 + 32 * k);
}
/* resolve_pending_foreign_call: n outputs; is_array[i]; lens[i] values each, concatenated big-endian */
int oracle_acvm_resolve(oracle_acvm_t *a, uint32_t n, const uint8_t *is_array, const uint32_t *lens,
                        const uint8_t *values_be32) {
    fc_result_t r;
    r.n = n;
    r.values = (fc_output_t *)calloc(n ? n : 1, sizeof(fc_output_t));
    size_t off = 0;
    for (uint32_t i = 0; i < n; i++) {
        r.values[i].is_array = is_array[i];
        if (is_array[i]) {
            r.values[i].n = lens[i];
            r.values[i].arr = (fr_t *)malloc((lens[i] + 1) * sizeof(fr_t));
            for (uint32_t k = 0; k < lens[i]; k++) fr_from_be_bytes_reduce(&r.values[i].arr[k], values_be32 + 32 * (off + k), 32);
            off += lens[i];
        } else {
            fr_from_be_bytes_reduce(&r.values[i].single, values_be32 + 32 * off, 32);
            off += 1;
        }
    }
    int rc = oracle_acvm_resolve_foreign_call(a, &r);
    for (uint32_t i = 0; i < n; i++) free(r.values[i].arr);
    free(r.values);
    return rc;
}

/* ---- batch: B independent instances of one circuit, in-order per instance, optional host threads ---- */
typedef struct {
    const circuit_t *c;
    const backend_t *be;
    size_t lo, hi, n_in;
    const uint32_t *ids;
    const uint8_t *values;
    acvm_result_t *results;
    uint8_t *assigned, *out_values;
    uint32_t nw;
    int mode;
} job_t;

static void *job_run(void *p) {
    job_t *j = (job_t *)p;
    oracle_inv_cache_t *cache = (j->mode & ORACLE_MODE_CACHE_INV) ? oracle_inv_cache_new(j->c) : NULL;
    oracle_acvm_t *a = NULL; /* one ACVM object per host thread, re-initialised between instances (pwg.c oracle_acvm_reset) */
    for (size_t i = j->lo; i < j->hi; i++) {
        if (!a) a = oracle_acvm_new_mode(j->c, j->be, j->n_in, j->ids, j->values + i * j->n_in * 32, j->mode, cache);
        else oracle_acvm_reset(a, j->n_in, j->ids, j->values + i * j->n_in * 32);
        oracle_acvm_solve(a);
        if (j->results) j->results[i] = a->res;
        if (j->assigned) {
            uint32_t n = a->nw < j->nw ? a->nw : j->nw;
            uint8_t *as = j->assigned + i * (size_t)j->nw;
            uint8_t *vs = j->out_values ? j->out_values + i * (size_t)j->nw * 32 : NULL;
            memset(as, 0, j->nw);
            for (uint32_t w = 0; w < n; w++) {
                as[w] = a->assigned[w];
                if (vs) {
                    if (a->assigned[w]) fr_to_be_bytes(&a->val[w], vs + 32 * (size_t)w);
                    else memset(vs + 32 * (size_t)w, 0, 32);
                }
            }
        }
    }
    oracle_acvm_free(a);
    oracle_inv_cache_free(cache);
    return NULL;
}

/* values_be32: [B][n_in][32]; results: [B] or NULL; assigned: [B][nw] or NULL; out_values: [B][nw][32] or NULL */
int oracle_solve_batch_mode(const circuit_t *c, int backend, size_t B, size_t n_in, const uint32_t *ids,
                            const uint8_t *values_be32, acvm_result_t *results, uint8_t *assigned, uint8_t *out_values,
                            uint32_t nw, int n_threads, int mode);
int oracle_solve_batch(const circuit_t *c, int backend, size_t B, size_t n_in, const uint32_t *ids,
                       const uint8_t *values_be32, acvm_result_t *results, uint8_t *assigned, uint8_t *out_values,
                       uint32_t nw, int n_threads) {
    return oracle_solve_batch_mode(c, backend, B, n_in, ids, values_be32, results, assigned, out_values, nw, n_threads, 0);
}
/* mode: ORACLE_MODE_* timing modes of the CPU baseline (pwg.h); results are identical in every mode */
int oracle_solve_batch_mode(const circuit_t *c, int backend, size_t B, size_t n_in, const uint32_t *ids,
                            const uint8_t *values_be32, acvm_result_t *results, uint8_t *assigned, uint8_t *out_values,
                            uint32_t nw, int n_threads, int mode) {
    if (n_threads < 1) n_threads = 1;
    if ((size_t)n_threads > B) n_threads = (int)(B ? B : 1);
    job_t *jobs = (job_t *)calloc((size_t)n_threads, sizeof(job_t));
    pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    for (int t = 0; t < n_threads; t++) {
        job_t *j = &jobs[t];
        j->c = c; j->be = oracle_backend(backend);
        j->lo = B * (size_t)t / (size_t)n_threads;
        j->hi = B * (size_t)(t + 1) / (size_t)n_threads;
        j->n_in = n_in; j->ids = ids; j->values = values_be32;
        j->results = results; j->assigned = assigned; j->out_values = out_values; j->nw = nw; j->mode = mode;
        if (n_threads == 1) job_run(j);
        else pthread_create(&th[t], NULL, job_run, j);
    }
    if (n_threads > 1)
        for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    free(jobs);
    free(th);
    return 0;
}

/* ---- field helpers for the Fr golden-vector tests ---- */
/* op: 0 add, 1 sub, 2 mul, 3 div, 4 neg(a), 5 inverse(a), 6 and(num_bits=aux), 7 xor(num_bits=aux) */
void oracle_fr_op(int op, const uint8_t a_be[32], const uint8_t b_be[32], uint32_t aux, uint8_t out_be[32]) {
    fr_t a, b, r;
    fr_from_be_bytes_reduce(&a, a_be, 32);
    fr_from_be_bytes_reduce(&b, b_be, 32);
    switch (op) {
    case 0: fr_add(&r, &a, &b); break;
    case 1: fr_sub(&r, &a, &b); break;
    case 2: fr_mul(&r, &a, &b); break;
    case 3: fr_div(&r, &a, &b); break;
    case 4: fr_neg(&r, &a); break;
    case 5: fr_inverse(&r, &a); break;
    case 6: fr_and_xor(&r, &a, &b, aux, 0); break;
    default: fr_and_xor(&r, &a, &b, aux, 1); break;
    }
    fr_to_be_bytes(&r, out_be);
}
uint32_t oracle_fr_num_bits(const uint8_t a_be[32]) {
    fr_t a;
    fr_from_be_bytes_reduce(&a, a_be, 32);
    return fr_num_bits(&a);
}
void oracle_fr_from_bytes_reduce(const uint8_t *bytes, size_t len, uint8_t out_be[32]) {
    fr_t a;
    fr_from_be_bytes_reduce(&a, bytes, len);
    fr_to_be_bytes(&a, out_be);
}
int oracle_fr_fetch_nearest_bytes(const uint8_t a_be[32], uint32_t num_bits, uint8_t out[32]) {
    fr_t a;
    fr_from_be_bytes_reduce(&a, a_be, 32);
    return fr_fetch_nearest_bytes(&a, num_bits, out);
}
