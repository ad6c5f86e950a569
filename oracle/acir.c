/*
 * oracle/acir.c -- TEST INFRASTRUCTURE ONLY (CPU oracle). Not part of the shipped product path.
 * bincode-1.3 (default config: fixint little-endian, u64 lengths, u32 enum tags, u8 Option tag)
 * reader for acir::circuit::Circuit. See acir.h for the reference anchors.
 */
#include "acir.h"
#include <stdlib.h>
#include <string.h>

/* ---- arena: every allocation of a circuit is released by acir_circuit_free ---- */
typedef struct block { struct block *next; } block_t;
typedef struct {
    const uint8_t *p, *end;
    int err;
    block_t *blocks;
    uint32_t max_w;
} rd_t;

static void *ralloc(rd_t *r, size_t n) {
    block_t *b = (block_t *)calloc(1, sizeof(block_t) + 16 + (n ? n : 1));
    if (!b) { r->err = 1; return NULL; }
    b->next = r->blocks;
    r->blocks = b;
    return (uint8_t *)b + 16 + (sizeof(block_t) > 16 ? sizeof(block_t) - 16 : 0);
}
static int need(rd_t *r, size_t n) {
    if (r->err || (size_t)(r->end - r->p) < n) { r->err = 1; return 0; }
    return 1;
}
static uint8_t rd_u8(rd_t *r) { if (!need(r, 1)) return 0; return *r->p++; }
static uint32_t rd_u32(rd_t *r) {
    if (!need(r, 4)) return 0;
    uint32_t v; memcpy(&v, r->p, 4); r->p += 4; return v;
}
static uint64_t rd_u64(rd_t *r) {
    if (!need(r, 8)) return 0;
    uint64_t v; memcpy(&v, r->p, 8); r->p += 8; return v;
}
/* length prefix, bounded by the remaining bytes so a corrupt length cannot exhaust memory */
static size_t rd_len(rd_t *r, size_t min_elem_bytes) {
    uint64_t n = rd_u64(r);
    if (r->err) return 0;
    if (min_elem_bytes && n > (uint64_t)(r->end - r->p) / min_elem_bytes) { r->err = 1; return 0; }
    return (size_t)n;
}
static uint32_t rd_witness(rd_t *r) {
    uint32_t w = rd_u32(r);
    if (w > r->max_w) r->max_w = w;
    return w;
}
static void rd_fr(rd_t *r, fr_t *out) {
    /* FieldElement = String of hex (generic_ark.rs:114-134): u64 len + bytes, from_hex reduces */
    size_t n = rd_len(r, 1);
    if (!need(r, n)) { fr_zero(out); return; }
    if (fr_from_hex(out, (const char *)r->p, n) != 0) r->err = 1;
    r->p += n;
}
static char *rd_string(rd_t *r) {
    size_t n = rd_len(r, 1);
    if (!need(r, n)) return NULL;
    char *s = (char *)ralloc(r, n + 1);
    if (!s) return NULL;
    memcpy(s, r->p, n);
    s[n] = 0;
    r->p += n;
    return s;
}
static void rd_expr(rd_t *r, expr_t *e) {
    e->n_mul = rd_len(r, 16);
    e->mul = (mul_term_t *)ralloc(r, e->n_mul * sizeof(mul_term_t));
    for (size_t i = 0; i < e->n_mul && !r->err; i++) {
        rd_fr(r, &e->mul[i].c);
        e->mul[i].l = rd_witness(r);
        e->mul[i].r = rd_witness(r);
    }
    e->n_lin = rd_len(r, 12);
    e->lin = (lin_term_t *)ralloc(r, e->n_lin * sizeof(lin_term_t));
    for (size_t i = 0; i < e->n_lin && !r->err; i++) {
        rd_fr(r, &e->lin[i].c);
        e->lin[i].w = rd_witness(r);
    }
    rd_fr(r, &e->qc);
}
static int rd_opt_expr(rd_t *r, expr_t *e) {
    uint8_t tag = rd_u8(r);
    if (tag == 0) return 0;
    if (tag != 1) { r->err = 1; return 0; }
    rd_expr(r, e);
    return 1;
}
static func_input_t rd_finput(rd_t *r) {
    func_input_t f;
    f.witness = rd_witness(r);
    f.num_bits = rd_u32(r);
    return f;
}
static void rd_finput_vec(rd_t *r, func_input_t **out, size_t *n) {
    *n = rd_len(r, 8);
    *out = (func_input_t *)ralloc(r, *n * sizeof(func_input_t));
    for (size_t i = 0; i < *n && !r->err; i++) (*out)[i] = rd_finput(r);
}
static void rd_finput_one(rd_t *r, func_input_t **out, size_t *n) {
    *n = 1;
    *out = (func_input_t *)ralloc(r, sizeof(func_input_t));
    if (*out) (*out)[0] = rd_finput(r);
}
static void rd_witness_vec(rd_t *r, uint32_t **out, size_t *n) {
    *n = rd_len(r, 4);
    *out = (uint32_t *)ralloc(r, *n * sizeof(uint32_t));
    for (size_t i = 0; i < *n && !r->err; i++) (*out)[i] = rd_witness(r);
}
static void rd_witness_n(rd_t *r, uint32_t **out, size_t *n, size_t count) {
    *n = count;
    *out = (uint32_t *)ralloc(r, count * sizeof(uint32_t));
    for (size_t i = 0; i < count && !r->err; i++) (*out)[i] = rd_witness(r);
}

static void rd_bb(rd_t *r, bb_call_t *b) {
    b->func = rd_u32(r);
    switch (b->func) {
    case BB_AND: case BB_XOR:
        rd_finput_one(r, &b->in[0], &b->n_in[0]);
        rd_finput_one(r, &b->in[1], &b->n_in[1]);
        rd_witness_n(r, &b->out, &b->n_out, 1);
        break;
    case BB_RANGE:
        rd_finput_one(r, &b->in[0], &b->n_in[0]);
        break;
    case BB_SHA256: case BB_BLAKE2S: case BB_KECCAK256:
        rd_finput_vec(r, &b->in[0], &b->n_in[0]);
        rd_witness_vec(r, &b->out, &b->n_out);
        break;
    case BB_SCHNORR_VERIFY:
        rd_finput_one(r, &b->in[0], &b->n_in[0]);
        rd_finput_one(r, &b->in[1], &b->n_in[1]);
        rd_finput_vec(r, &b->in[2], &b->n_in[2]);
        rd_finput_vec(r, &b->in[3], &b->n_in[3]);
        rd_witness_n(r, &b->out, &b->n_out, 1);
        break;
    case BB_PEDERSEN:
        rd_finput_vec(r, &b->in[0], &b->n_in[0]);
        b->domain_separator = rd_u32(r);
        rd_witness_n(r, &b->out, &b->n_out, 2);
        break;
    case BB_HASH_TO_FIELD_128:
        rd_finput_vec(r, &b->in[0], &b->n_in[0]);
        rd_witness_n(r, &b->out, &b->n_out, 1);
        break;
    case BB_ECDSA_SECP256K1: case BB_ECDSA_SECP256R1:
        for (int g = 0; g < 4; g++) rd_finput_vec(r, &b->in[g], &b->n_in[g]);
        rd_witness_n(r, &b->out, &b->n_out, 1);
        break;
    case BB_FIXED_BASE_SCALAR_MUL:
        rd_finput_one(r, &b->in[0], &b->n_in[0]);
        rd_finput_one(r, &b->in[1], &b->n_in[1]);
        rd_witness_n(r, &b->out, &b->n_out, 2);
        break;
    case BB_KECCAK256_VAR:
        rd_finput_vec(r, &b->in[0], &b->n_in[0]);
        rd_finput_one(r, &b->in[1], &b->n_in[1]);
        rd_witness_vec(r, &b->out, &b->n_out);
        break;
    case BB_RECURSIVE_AGGREGATION: {
        rd_finput_vec(r, &b->in[0], &b->n_in[0]);
        rd_finput_vec(r, &b->in[1], &b->n_in[1]);
        rd_finput_vec(r, &b->in[2], &b->n_in[2]);
        rd_finput_one(r, &b->in[3], &b->n_in[3]);
        uint8_t tag = rd_u8(r);
        if (tag == 1) { b->has_in_agg = 1; rd_finput_vec(r, &b->in_agg, &b->n_in_agg); }
        else if (tag != 0) r->err = 1;
        rd_witness_vec(r, &b->out, &b->n_out);
        break;
    }
    default: r->err = 1;
    }
}

static void rd_directive(rd_t *r, directive_t *d) {
    d->kind = rd_u32(r);
    switch (d->kind) {
    case DIR_QUOTIENT:
        rd_expr(r, &d->a);
        rd_expr(r, &d->b);
        d->q = rd_witness(r);
        d->r = rd_witness(r);
        d->has_predicate = rd_opt_expr(r, &d->predicate);
        break;
    case DIR_TO_LE_RADIX:
        rd_expr(r, &d->a);
        rd_witness_vec(r, &d->bw, &d->n_bw);
        d->radix = rd_u32(r);
        break;
    case DIR_PERMUTATION_SORT: {
        d->n_sort_inputs = rd_len(r, 8);
        d->sort_inputs = (expr_t **)ralloc(r, d->n_sort_inputs * sizeof(expr_t *));
        d->sort_input_len = (size_t *)ralloc(r, d->n_sort_inputs * sizeof(size_t));
        for (size_t i = 0; i < d->n_sort_inputs && !r->err; i++) {
            size_t n = rd_len(r, 24);
            d->sort_input_len[i] = n;
            d->sort_inputs[i] = (expr_t *)ralloc(r, n * sizeof(expr_t));
            for (size_t j = 0; j < n && !r->err; j++) rd_expr(r, &d->sort_inputs[i][j]);
        }
        d->tuple = rd_u32(r);
        rd_witness_vec(r, &d->bw, &d->n_bw);
        d->n_sort_by = rd_len(r, 4);
        d->sort_by = (uint32_t *)ralloc(r, d->n_sort_by * sizeof(uint32_t));
        for (size_t i = 0; i < d->n_sort_by && !r->err; i++) d->sort_by[i] = rd_u32(r);
        break;
    }
    default: r->err = 1;
    }
}

static reg_or_mem_t rd_rom(rd_t *r) {
    reg_or_mem_t m;
    memset(&m, 0, sizeof m);
    m.kind = rd_u32(r);
    m.reg = rd_u64(r);
    if (m.kind == ROM_HEAP_ARRAY || m.kind == ROM_HEAP_VECTOR) m.size = rd_u64(r);
    else if (m.kind != ROM_REGISTER) r->err = 1;
    return m;
}
static void rd_rom_vec(rd_t *r, reg_or_mem_t **out, size_t *n) {
    *n = rd_len(r, 12);
    *out = (reg_or_mem_t *)ralloc(r, *n * sizeof(reg_or_mem_t));
    for (size_t i = 0; i < *n && !r->err; i++) (*out)[i] = rd_rom(r);
}

static void rd_brillig_op(rd_t *r, brillig_op_t *o) {
    o->op = rd_u32(r);
    switch (o->op) {
    case BR_BINARY_FIELD_OP:
        o->a = rd_u64(r); o->sub_op = rd_u32(r); o->b = rd_u64(r); o->c = rd_u64(r);
        if (o->sub_op > BF_EQUALS) r->err = 1;
        break;
    case BR_BINARY_INT_OP:
        o->a = rd_u64(r); o->sub_op = rd_u32(r); o->bit_size = rd_u32(r); o->b = rd_u64(r); o->c = rd_u64(r);
        if (o->sub_op > BI_SHR) r->err = 1;
        break;
    case BR_JUMP_IF_NOT: case BR_JUMP_IF:
        o->a = rd_u64(r); o->location = rd_u64(r);
        break;
    case BR_JUMP: case BR_CALL:
        o->location = rd_u64(r);
        break;
    case BR_CONST:
        o->a = rd_u64(r); rd_fr(r, &o->value);
        break;
    case BR_RETURN: case BR_TRAP: case BR_STOP:
        break;
    case BR_FOREIGN_CALL:
        o->function = rd_string(r);
        rd_rom_vec(r, &o->dests, &o->n_dests);
        rd_rom_vec(r, &o->inputs, &o->n_inputs);
        break;
    case BR_MOV: case BR_LOAD: case BR_STORE:
        o->a = rd_u64(r); o->b = rd_u64(r);
        break;
    case BR_BLACK_BOX: {
        o->bbop = rd_u32(r);
        static const int nwords[9] = {4, 4, 4, 3, 9, 9, 7, 5, 4};
        if (o->bbop > BBOP_FIXED_BASE) { r->err = 1; break; }
        for (int i = 0; i < nwords[o->bbop]; i++) o->bb[i] = rd_u64(r);
        break;
    }
    default: r->err = 1;
    }
}

static void rd_brillig(rd_t *r, brillig_t *b) {
    b->n_inputs = rd_len(r, 4);
    b->inputs = (brillig_input_t *)ralloc(r, b->n_inputs * sizeof(brillig_input_t));
    for (size_t i = 0; i < b->n_inputs && !r->err; i++) {
        uint32_t tag = rd_u32(r);
        if (tag == 0) rd_expr(r, &b->inputs[i].single);
        else if (tag == 1) {
            b->inputs[i].is_array = 1;
            b->inputs[i].n = rd_len(r, 24);
            b->inputs[i].arr = (expr_t *)ralloc(r, b->inputs[i].n * sizeof(expr_t));
            for (size_t j = 0; j < b->inputs[i].n && !r->err; j++) rd_expr(r, &b->inputs[i].arr[j]);
        } else r->err = 1;
    }
    b->n_outputs = rd_len(r, 4);
    b->outputs = (brillig_output_t *)ralloc(r, b->n_outputs * sizeof(brillig_output_t));
    for (size_t i = 0; i < b->n_outputs && !r->err; i++) {
        uint32_t tag = rd_u32(r);
        if (tag == 0) b->outputs[i].w = rd_witness(r);
        else if (tag == 1) {
            b->outputs[i].is_array = 1;
            rd_witness_vec(r, &b->outputs[i].arr, &b->outputs[i].n);
        } else r->err = 1;
    }
    b->n_fc_results = rd_len(r, 8);
    b->cap_fc_results = b->n_fc_results;
    b->fc_results = (fc_result_t *)ralloc(r, b->n_fc_results * sizeof(fc_result_t));
    for (size_t i = 0; i < b->n_fc_results && !r->err; i++) {
        fc_result_t *f = &b->fc_results[i];
        f->n = rd_len(r, 4);
        f->values = (fc_output_t *)ralloc(r, f->n * sizeof(fc_output_t));
        for (size_t j = 0; j < f->n && !r->err; j++) {
            uint32_t tag = rd_u32(r);
            if (tag == 0) rd_fr(r, &f->values[j].single);
            else if (tag == 1) {
                f->values[j].is_array = 1;
                f->values[j].n = rd_len(r, 8);
                f->values[j].arr = (fr_t *)ralloc(r, f->values[j].n * sizeof(fr_t));
                for (size_t k = 0; k < f->values[j].n && !r->err; k++) rd_fr(r, &f->values[j].arr[k]);
            } else r->err = 1;
        }
    }
    b->n_bytecode = rd_len(r, 4);
    b->bytecode = (brillig_op_t *)ralloc(r, b->n_bytecode * sizeof(brillig_op_t));
    for (size_t i = 0; i < b->n_bytecode && !r->err; i++) rd_brillig_op(r, &b->bytecode[i]);
    b->has_predicate = rd_opt_expr(r, &b->predicate);
}

static void rd_opcode(rd_t *r, opcode_t *o) {
    o->kind = rd_u32(r);
    switch (o->kind) {
    case OP_ARITHMETIC: rd_expr(r, &o->expr); break;
    case OP_BLACKBOX: rd_bb(r, &o->bb); break;
    case OP_DIRECTIVE: rd_directive(r, &o->dir); break;
    case OP_BRILLIG: rd_brillig(r, &o->brillig); break;
    case OP_MEMORY_OP:
        o->block_id = rd_u32(r);
        rd_expr(r, &o->mem_operation);
        rd_expr(r, &o->mem_index);
        rd_expr(r, &o->mem_value);
        o->has_predicate = rd_opt_expr(r, &o->predicate);
        break;
    case OP_MEMORY_INIT:
        o->block_id = rd_u32(r);
        rd_witness_vec(r, &o->init, &o->n_init);
        break;
    default: r->err = 1;
    }
}

circuit_t *acir_circuit_parse(const uint8_t *buf, size_t len) {
    rd_t r;
    memset(&r, 0, sizeof r);
    r.p = buf;
    r.end = buf + len;
    circuit_t *c = (circuit_t *)ralloc(&r, sizeof(circuit_t) + sizeof(block_t *));
    if (!c) return NULL;
    c->current_witness_index = rd_u32(&r);
    c->n_opcodes = rd_len(&r, 4);
    c->opcodes = (opcode_t *)ralloc(&r, c->n_opcodes * sizeof(opcode_t));
    for (size_t i = 0; i < c->n_opcodes && !r.err; i++) rd_opcode(&r, &c->opcodes[i]);
    rd_witness_vec(&r, &c->private_parameters, &c->n_private);
    rd_witness_vec(&r, &c->public_parameters, &c->n_public);
    rd_witness_vec(&r, &c->return_values, &c->n_return);
    c->n_assert = rd_len(&r, 12);
    c->assert_messages = (assert_msg_t *)ralloc(&r, c->n_assert * sizeof(assert_msg_t));
    for (size_t i = 0; i < c->n_assert && !r.err; i++) {
        assert_msg_t *m = &c->assert_messages[i];
        uint32_t tag = rd_u32(&r);
        if (tag == 0) m->acir_index = rd_u64(&r);
        else if (tag == 1) { m->is_brillig = 1; m->acir_index = rd_u64(&r); m->brillig_index = rd_u64(&r); }
        else r.err = 1;
        m->message = rd_string(&r);
    }
    if (!r.err && r.p != r.end) r.err = 1; /* trailing bytes */
    c->max_witness = r.max_w > c->current_witness_index ? r.max_w : c->current_witness_index;
    /* stash the block list head just behind the struct so free can find it */
    *(block_t **)((uint8_t *)c + sizeof(circuit_t)) = r.blocks;
    if (r.err) { acir_circuit_free(c); return NULL; }
    return c;
}

void acir_circuit_free(circuit_t *c) {
    if (!c) return;
    block_t *b = *(block_t **)((uint8_t *)c + sizeof(circuit_t));
    while (b) { block_t *n = b->next; free(b); b = n; }
}

void expr_free(expr_t *e) { free(e->mul); free(e->lin); e->mul = NULL; e->lin = NULL; e->n_mul = e->n_lin = 0; }
void expr_clone(expr_t *dst, const expr_t *src) {
    dst->n_mul = src->n_mul; dst->n_lin = src->n_lin; dst->qc = src->qc;
    dst->mul = (mul_term_t *)malloc((src->n_mul ? src->n_mul : 1) * sizeof(mul_term_t));
    dst->lin = (lin_term_t *)malloc((src->n_lin ? src->n_lin : 1) * sizeof(lin_term_t));
    memcpy(dst->mul, src->mul, src->n_mul * sizeof(mul_term_t));
    memcpy(dst->lin, src->lin, src->n_lin * sizeof(lin_term_t));
}
