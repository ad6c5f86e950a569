/*
 * oracle/brillig_vm.c -- TEST INFRASTRUCTURE ONLY (CPU oracle). Not part of the shipped product path.
 *
 * Restates the Brillig register VM and its ACVM glue:
 *   /root/reference/brillig_vm/src/lib.rs:61-390        VM, process_opcode, foreign-call replay
 *   /root/reference/brillig_vm/src/arithmetic.rs:7-98   field / fixed-width integer ALU (BigUint semantics)
 *   /root/reference/brillig_vm/src/registers.rs:4-43    unset register reads 0, max 2^16
 *   /root/reference/brillig_vm/src/memory.rs:4-45       reads out of range panic, writes grow with 0
 *   /root/reference/brillig_vm/src/black_box.rs:42-165  black box ops
 *   /root/reference/acvm/src/pwg/brillig.rs:20-150      BrilligSolver::solve, zero_out_brillig_outputs
 * Integer ops take any bit_size like the reference's BigUint arithmetic (operands are field elements < p < 2^254, so past 256 bits only
 * Sub and Mul can still see the modulus 2^bit_size; see int_op).
 * Pinned by tests/test_oracle_acvm.py (the Brillig tests there) against brillig_vm/src/arithmetic.rs:149-234 known answers,
 * acvm/tests/solver.rs:308-608 and acvm_js/test/shared/{foreign_call,complex_foreign_call}.ts.
 */
#include "hashes.h"
#include "pwg.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAX_REGISTERS 65536
typedef unsigned __int128 u128;

typedef struct {
    fr_t *reg; size_t n_reg;
    fr_t *mem; size_t n_mem;
    size_t pc, fc_counter;
    size_t *call_stack; size_t n_cs, cap_cs;
    const brillig_t *b;
    oracle_acvm_t *a;
    size_t acir_index;
    int status; /* 0 in progress, 1 finished, 2 failure, 3 foreign call wait, 4 panic */
    char msg[200];
} vm_t;

static fr_t reg_get(vm_t *vm, uint64_t i) {
    fr_t z;
    fr_zero(&z);
    if (i >= MAX_REGISTERS) { vm->status = 4; snprintf(vm->msg, sizeof vm->msg, "Reading register past maximum!"); return z; }
    return i < vm->n_reg ? vm->reg[i] : z;
}
static void reg_set(vm_t *vm, uint64_t i, const fr_t *v) {
    if (i >= MAX_REGISTERS) { vm->status = 4; snprintf(vm->msg, sizeof vm->msg, "Writing register past maximum!"); return; }
    if (i >= vm->n_reg) {
        vm->reg = (fr_t *)realloc(vm->reg, (i + 1) * sizeof(fr_t));
        memset(vm->reg + vm->n_reg, 0, (i + 1 - vm->n_reg) * sizeof(fr_t));
        vm->n_reg = i + 1;
    }
    vm->reg[i] = *v;
}
/* Value::to_usize (brillig/src/value.rs:46-49): panics above u64 */
static int to_usize(vm_t *vm, const fr_t *v, uint64_t *out) {
    if (!fr_try_to_u64(v, out)) { vm->status = 4; snprintf(vm->msg, sizeof vm->msg, "register does not fit into u64"); return 1; }
    return 0;
}
static int mem_read_slice(vm_t *vm, uint64_t ptr, uint64_t len, const fr_t **out) {
    if (ptr > vm->n_mem || len > vm->n_mem - ptr) { vm->status = 4; snprintf(vm->msg, sizeof vm->msg, "memory read out of range"); return 1; }
    *out = vm->mem + ptr;
    return 0;
}
static void mem_write_slice(vm_t *vm, uint64_t ptr, const fr_t *vals, uint64_t n) {
    if (ptr + n > (1ULL << 28)) { vm->status = 4; snprintf(vm->msg, sizeof vm->msg, "memory write beyond oracle limit"); return; }
    if (ptr + n > vm->n_mem) {
        vm->mem = (fr_t *)realloc(vm->mem, (ptr + n) * sizeof(fr_t));
        memset(vm->mem + vm->n_mem, 0, (ptr + n - vm->n_mem) * sizeof(fr_t));
        vm->n_mem = ptr + n;
    }
    memcpy(vm->mem + ptr, vals, n * sizeof(fr_t));
}

/* ---- 256/512-bit helpers on canonical integers ---- */
static void mask_bits(uint64_t *v, int limbs, uint32_t bits) {
    for (int i = 0; i < limbs; i++) {
        uint32_t lo = 64u * (uint32_t)i;
        if (bits <= lo) v[i] = 0;
        else if (bits < lo + 64) v[i] &= (1ULL << (bits - lo)) - 1;
    }
}
static int cmp4(const uint64_t a[4], const uint64_t b[4]) {
    for (int i = 3; i >= 0; i--) {
        if (a[i] < b[i]) return -1;
        if (a[i] > b[i]) return 1;
    }
    return 0;
}
static int is_zero4(const uint64_t a[4]) { return !(a[0] | a[1] | a[2] | a[3]); }
static void divrem4(const uint64_t a[4], const uint64_t b[4], uint64_t q[4], uint64_t r[4]) {
    memset(q, 0, 32);
    memset(r, 0, 32);
    for (int i = 255; i >= 0; i--) {
        r[3] = (r[3] << 1) | (r[2] >> 63);
        r[2] = (r[2] << 1) | (r[1] >> 63);
        r[1] = (r[1] << 1) | (r[0] >> 63);
        r[0] = (r[0] << 1) | ((a[i / 64] >> (i % 64)) & 1);
        if (cmp4(r, b) >= 0) {
            uint64_t borrow = 0;
            for (int k = 0; k < 4; k++) {
                u128 d = (u128)r[k] - b[k] - borrow;
                r[k] = (uint64_t)d;
                borrow = (uint64_t)(d >> 64) & 1;
            }
            q[i / 64] |= 1ULL << (i % 64);
        }
    }
}
static void from_limbs_reduce(fr_t *out, const uint64_t *v, int limbs) {
    uint8_t be[64];
    for (int i = 0; i < limbs * 8; i++) be[limbs * 8 - 1 - i] = (uint8_t)(v[i / 8] >> (8 * (i % 8)));
    fr_from_be_bytes_reduce(out, be, (size_t)limbs * 8);
}
/* two's complement view used by SignedDiv (arithmetic.rs:84-98): value = a if a < 2^(bits-1) else a - 2^bits.
 * returns sign (0 / 1) and magnitude */
static int to_signed(const uint64_t a[4], uint32_t bits, uint64_t mag[4]) {
    uint64_t half[4] = {0, 0, 0, 0}, full[5] = {0, 0, 0, 0, 0};
    half[(bits - 1) / 64] = 1ULL << ((bits - 1) % 64);
    if (cmp4(a, half) < 0) { memcpy(mag, a, 32); return 0; }
    full[bits / 64] = 1ULL << (bits % 64); /* 2^bits, bits <= 256 */
    /* a - 2^bits : negative iff a < 2^bits */
    uint64_t a5[5] = {a[0], a[1], a[2], a[3], 0};
    int lt = 0;
    for (int i = 4; i >= 0; i--) { if (a5[i] < full[i]) { lt = 1; break; } if (a5[i] > full[i]) break; }
    uint64_t borrow = 0, res[5];
    const uint64_t *x = lt ? full : a5, *y = lt ? a5 : full;
    for (int k = 0; k < 5; k++) {
        u128 d = (u128)x[k] - y[k] - borrow;
        res[k] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
    memcpy(mag, res, 32); /* |a - 2^bits| < 2^256 */
    return lt;
}

/* 2^bits mod p for any u32 bits (square and multiply) */
static void fr_pow2_bits(fr_t *out, uint32_t bits) {
    fr_t base, acc;
    fr_from_u64(&base, 2);
    fr_from_u64(&acc, 1);
    for (uint32_t e = bits; e; e >>= 1) {
        if (e & 1) fr_mul(&acc, &acc, &base);
        fr_mul(&base, &base, &base);
    }
    *out = acc;
}

static void vm_panic(vm_t *vm, const char *m) { vm->status = 4; snprintf(vm->msg, sizeof vm->msg, "%s", m); }

/* arithmetic.rs:23-81 evaluate_binary_bigint_op */
static void int_op(vm_t *vm, uint32_t op, uint32_t bits, const fr_t *fa, const fr_t *fb, fr_t *out) {
    uint64_t a[4], b[4], r8[8];
    fr_to_canonical(fa, a);
    fr_to_canonical(fb, b);
    memset(r8, 0, sizeof r8);
    /* bit_size > 256 (the reference's BigUint takes any size; operands are < p < 2^254): Add cannot wrap; the masks of the other ops
     * are no-ops on 256-bit values; Mul masks its 512-bit product; Sub with a < b is 2^bits + a - b, a number of `bits` bits that only its
     * residue mod p survives (from_be_bytes_reduce): (2^bits mod p) + a - b in the field; SignedDiv sees two non-negative numbers. */
    switch (op) {
    case BI_ADD: {
        u128 c = 0;
        for (int i = 0; i < 4; i++) { c += (u128)a[i] + b[i]; r8[i] = (uint64_t)c; c >>= 64; }
        r8[4] = (uint64_t)c;
        mask_bits(r8, 8, bits);
        break;
    }
    case BI_SUB: { /* (2^bits + a - b) % 2^bits ; BigUint underflow panics when b > 2^bits + a */
        if (bits > 256) {
            fr_t d;
            fr_sub(&d, fa, fb);
            if (cmp4(a, b) >= 0) { *out = d; return; }
            fr_t p2;
            fr_pow2_bits(&p2, bits);
            fr_add(out, &p2, &d);
            return;
        }
        uint64_t t[5] = {a[0], a[1], a[2], a[3], 0}, bb[5] = {b[0], b[1], b[2], b[3], 0};
        u128 c = (u128)t[bits / 64] + (1ULL << (bits % 64));
        t[bits / 64] = (uint64_t)c;
        for (uint32_t i = bits / 64 + 1; i < 5 && (c >> 64); i++) { c = (u128)t[i] + 1; t[i] = (uint64_t)c; }
        uint64_t borrow = 0;
        for (int k = 0; k < 5; k++) {
            u128 d = (u128)t[k] - bb[k] - borrow;
            r8[k] = (uint64_t)d;
            borrow = (uint64_t)(d >> 64) & 1;
        }
        if (borrow) { vm_panic(vm, "attempt to subtract with overflow"); return; }
        mask_bits(r8, 8, bits);
        break;
    }
    case BI_MUL:
        for (int i = 0; i < 4; i++) {
            u128 c = 0;
            for (int j = 0; j < 4; j++) { c += (u128)a[i] * b[j] + r8[i + j]; r8[i + j] = (uint64_t)c; c >>= 64; }
            r8[i + 4] = (uint64_t)c;
        }
        mask_bits(r8, 8, bits);
        break;
    case BI_UNSIGNED_DIV: {
        uint64_t q[4], r[4];
        mask_bits(a, 4, bits);
        mask_bits(b, 4, bits);
        if (is_zero4(b)) { vm_panic(vm, "attempt to divide by zero"); return; }
        divrem4(a, b, q, r);
        memcpy(r8, q, 32);
        break;
    }
    case BI_SIGNED_DIV: {
        if (bits == 0) { vm_panic(vm, "attempt to subtract with overflow"); return; }
        uint64_t ma[4], mb[4], q[4], r[4];
        int sa = 0, sb = 0;
        if (bits > 256) { memcpy(ma, a, 32); memcpy(mb, b, 32); } /* a, b < 2^254 <= 2^(bits-1): both non-negative */
        else { sa = to_signed(a, bits, ma); sb = to_signed(b, bits, mb); }
        if (is_zero4(mb)) { vm_panic(vm, "attempt to divide by zero"); return; }
        divrem4(ma, mb, q, r); /* BigInt division truncates toward zero */
        int neg = (sa ^ sb) && !is_zero4(q);
        if (!neg) memcpy(r8, q, 32);
        else { /* 2^bits - |q| ; BigUint underflow panics if |q| > 2^bits */
            uint64_t full[5] = {0, 0, 0, 0, 0}, q5[5] = {q[0], q[1], q[2], q[3], 0};
            full[bits / 64] = 1ULL << (bits % 64);
            uint64_t borrow = 0;
            for (int k = 0; k < 5; k++) {
                u128 d = (u128)full[k] - q5[k] - borrow;
                r8[k] = (uint64_t)d;
                borrow = (uint64_t)(d >> 64) & 1;
            }
            if (borrow) { vm_panic(vm, "attempt to subtract with overflow"); return; }
        }
        break;
    }
    case BI_EQUALS: case BI_LT: case BI_LTE: {
        mask_bits(a, 4, bits);
        mask_bits(b, 4, bits);
        int c = cmp4(a, b);
        r8[0] = op == BI_EQUALS ? c == 0 : op == BI_LT ? c < 0 : c <= 0;
        break;
    }
    case BI_AND: for (int i = 0; i < 4; i++) r8[i] = a[i] & b[i]; mask_bits(r8, 8, bits); break;
    case BI_OR: for (int i = 0; i < 4; i++) r8[i] = a[i] | b[i]; mask_bits(r8, 8, bits); break;
    case BI_XOR: for (int i = 0; i < 4; i++) r8[i] = a[i] ^ b[i]; mask_bits(r8, 8, bits); break;
    case BI_SHL: case BI_SHR: {
        if (bits > 128) { vm_panic(vm, "unsupported bit size for right shift"); return; }
        if (b[2] | b[3]) { vm_panic(vm, "called `Option::unwrap()` on a `None` value"); return; } /* to_u128().unwrap() */
        if (op == BI_SHL) {
            if (!b[1] && b[0] < 256) {
                uint32_t s = (uint32_t)b[0];
                for (int i = 7; i >= 0; i--) {
                    int src = i - (int)(s / 64);
                    uint64_t v = 0;
                    if (src >= 0 && src < 4) v = a[src] << (s % 64);
                    if ((s % 64) && src - 1 >= 0 && src - 1 < 4) v |= a[src - 1] >> (64 - s % 64);
                    r8[i] = v;
                }
                mask_bits(r8, 8, bits);
            } /* else a << b has >= 256 low zero bits; mod 2^bits (bits <= 128) is 0 */
        } else {
            if (!b[1] && b[0] < 256) {
                uint32_t s = (uint32_t)b[0];
                for (int i = 0; i < 4; i++) {
                    uint32_t src = (uint32_t)i + s / 64;
                    uint64_t v = 0;
                    if (src < 4) v = a[src] >> (s % 64);
                    if ((s % 64) && src + 1 < 4) v |= a[src + 1] << (64 - s % 64);
                    r8[i] = v;
                }
                mask_bits(r8, 8, bits);
            }
        }
        break;
    }
    default: vm_panic(vm, "bad int op"); return;
    }
    from_limbs_reduce(out, r8, 8);
}

static int read_u8_vec(vm_t *vm, uint64_t ptr, uint64_t len, uint8_t **out) {
    const fr_t *s;
    if (mem_read_slice(vm, ptr, len, &s)) return 1;
    uint8_t *m = (uint8_t *)malloc(len + 1);
    for (uint64_t i = 0; i < len; i++) { uint8_t be[32]; fr_to_be_bytes(&s[i], be); m[i] = be[31]; }
    *out = m;
    return 0;
}
static int heap_vector(vm_t *vm, uint64_t preg, uint64_t sreg, uint64_t *ptr, uint64_t *len) {
    fr_t p = reg_get(vm, preg), s = reg_get(vm, sreg);
    return vm->status == 4 || to_usize(vm, &p, ptr) || to_usize(vm, &s, len);
}
static int reg_usize(vm_t *vm, uint64_t r, uint64_t *out) {
    fr_t p = reg_get(vm, r);
    return vm->status == 4 || to_usize(vm, &p, out);
}

/* black_box.rs:42-165. returns 0 ok, 1 BlackBoxResolutionError (vm fails with its Display string) */
static int vm_black_box(vm_t *vm, const brillig_op_t *o) {
    const backend_t *be = vm->a->backend;
    char err[160];
    err[0] = 0;
    uint64_t ptr, len, optr;
    switch (o->bbop) {
    case BBOP_SHA256: case BBOP_BLAKE2S: case BBOP_KECCAK256: {
        uint8_t *m, d[32];
        fr_t vals[32];
        if (heap_vector(vm, o->bb[0], o->bb[1], &ptr, &len) || read_u8_vec(vm, ptr, len, &m)) return 0;
        if (o->bbop == BBOP_SHA256) oracle_sha256(m, len, d);
        else if (o->bbop == BBOP_BLAKE2S) oracle_blake2s(m, len, d);
        else oracle_keccak256(m, len, d);
        free(m);
        for (int i = 0; i < 32; i++) fr_from_u64(&vals[i], d[i]);
        if (reg_usize(vm, o->bb[2], &optr)) return 0;
        mem_write_slice(vm, optr, vals, 32);
        return 0;
    }
    case BBOP_HASH_TO_FIELD: {
        uint8_t *m, d[32];
        fr_t f;
        if (heap_vector(vm, o->bb[0], o->bb[1], &ptr, &len) || read_u8_vec(vm, ptr, len, &m)) return 0;
        oracle_blake2s(m, len, d);
        free(m);
        fr_from_be_bytes_reduce(&f, d, 32);
        reg_set(vm, o->bb[2], &f);
        return 0;
    }
    case BBOP_ECDSA_K1: case BBOP_ECDSA_R1: { /* black_box.rs:74-131: hashed_msg vector, pkx / pky / signature arrays, result */
        const char *fname = o->bbop == BBOP_ECDSA_K1 ? "ecdsa_secp256k1" : "ecdsa_secp256r1";
        static const char *what[3] = {"Invalid public key x length", "Invalid public key y length", "Invalid signature length"};
        static const uint64_t want[3] = {32, 32, 64};
        uint8_t *arr[3] = {0, 0, 0};
        for (int g = 0; g < 3; g++) {
            uint64_t ap;
            if (reg_usize(vm, o->bb[2 + 2 * g], &ap) || read_u8_vec(vm, ap, o->bb[3 + 2 * g], &arr[g])) { for (int k = 0; k < g; k++) free(arr[k]); return 0; }
            if (o->bb[3 + 2 * g] != want[g]) {
                snprintf(vm->msg, sizeof vm->msg, "failed to solve blackbox function: %s, reason: %s", fname, what[g]);
                for (int k = 0; k <= g; k++) free(arr[k]);
                return 1;
            }
        }
        uint8_t *m;
        if (heap_vector(vm, o->bb[0], o->bb[1], &ptr, &len) || read_u8_vec(vm, ptr, len, &m)) { for (int k = 0; k < 3; k++) free(arr[k]); return 0; }
        int v = oracle_ecdsa_verify(o->bbop == BBOP_ECDSA_R1, m, len, arr[0], arr[1], arr[2]);
        free(m);
        for (int k = 0; k < 3; k++) free(arr[k]);
        if (v < 0) { vm_panic(vm, oracle_ecdsa_panic_text(v)); return 0; }
        fr_t r;
        fr_from_u64(&r, (uint64_t)v);
        reg_set(vm, o->bb[8], &r);
        return 0;
    }
    case BBOP_SCHNORR: {
        fr_t pkx = reg_get(vm, o->bb[0]), pky = reg_get(vm, o->bb[1]);
        uint8_t *msg, *sig;
        uint64_t mp, ml, sp, sl;
        if (heap_vector(vm, o->bb[2], o->bb[3], &mp, &ml) || read_u8_vec(vm, mp, ml, &msg)) return 0;
        if (heap_vector(vm, o->bb[4], o->bb[5], &sp, &sl) || read_u8_vec(vm, sp, sl, &sig)) { free(msg); return 0; }
        int ok = 0;
        int rc = be->schnorr_verify(be->ctx, &pkx, &pky, sig, sl, msg, ml, &ok, err, sizeof err);
        free(msg); free(sig);
        if (rc == 3) { vm_panic(vm, err); return 0; }
        if (rc) { snprintf(vm->msg, sizeof vm->msg, rc == 2 ? "unsupported blackbox function: schnorr_verify" : "failed to solve blackbox function: schnorr_verify, reason: %s", err); return 1; }
        fr_t v;
        fr_from_u64(&v, ok ? 1 : 0);
        reg_set(vm, o->bb[6], &v);
        return 0;
    }
    case BBOP_PEDERSEN: {
        const fr_t *in;
        if (heap_vector(vm, o->bb[0], o->bb[1], &ptr, &len) || mem_read_slice(vm, ptr, len, &in)) return 0;
        fr_t ds = reg_get(vm, o->bb[2]);
        uint64_t lo, hi;
        fr_to_u128(&ds, &lo, &hi);
        if (hi || lo > 0xffffffffULL) {
            snprintf(vm->msg, sizeof vm->msg, "failed to solve blackbox function: pedersen, reason: Invalid signature length");
            return 1;
        }
        fr_t xy[2];
        int rc = be->pedersen(be->ctx, in, len, (uint32_t)lo, &xy[0], &xy[1], err, sizeof err);
        if (rc == 3) { vm_panic(vm, err); return 0; }
        if (rc) { snprintf(vm->msg, sizeof vm->msg, rc == 2 ? "unsupported blackbox function: pedersen" : "failed to solve blackbox function: pedersen, reason: %s", err); return 1; }
        if (reg_usize(vm, o->bb[3], &optr)) return 0;
        mem_write_slice(vm, optr, xy, 2);
        return 0;
    }
    case BBOP_FIXED_BASE: {
        fr_t lo = reg_get(vm, o->bb[0]), hi = reg_get(vm, o->bb[1]), xy[2];
        int rc = be->fixed_base_scalar_mul(be->ctx, &lo, &hi, &xy[0], &xy[1], err, sizeof err);
        if (rc == 3) { vm_panic(vm, err); return 0; }
        if (rc) { snprintf(vm->msg, sizeof vm->msg, rc == 2 ? "unsupported blackbox function: fixed_base_scalar_mul" : "failed to solve blackbox function: fixed_base_scalar_mul, reason: %s", err); return 1; }
        if (reg_usize(vm, o->bb[2], &optr)) return 0;
        mem_write_slice(vm, optr, xy, 2);
        return 0;
    }
    }
    vm_panic(vm, "bad black box op");
    return 0;
}

static const fc_result_t *fc_result_at(vm_t *vm, size_t i, size_t *total) {
    size_t n0 = vm->b->n_fc_results, n1 = vm->a->n_extra_fc[vm->acir_index];
    *total = n0 + n1;
    if (i < n0) return &vm->b->fc_results[i];
    if (i < n0 + n1) return &vm->a->extra_fc[vm->acir_index][i - n0];
    return NULL;
}

static void vm_set_pc(vm_t *vm, size_t v) { /* lib.rs:322-329 */
    vm->pc = v;
    if (vm->pc >= vm->b->n_bytecode) vm->status = 1;
}
static void vm_fail(vm_t *vm, const char *m) { /* lib.rs:127-133 */
    vm->status = 2;
    if (m != vm->msg) snprintf(vm->msg, sizeof vm->msg, "%s", m);
}

static void vm_step(vm_t *vm) {
    if (vm->pc >= vm->b->n_bytecode) { vm_panic(vm, "index out of bounds: bytecode"); return; }
    const brillig_op_t *o = &vm->b->bytecode[vm->pc];
    fr_t x, y, r;
    uint64_t u;
    switch (o->op) {
    case BR_BINARY_FIELD_OP: /* arithmetic.rs:7-20 */
        x = reg_get(vm, o->b);
        y = reg_get(vm, o->c);
        switch (o->sub_op) {
        case BF_ADD: fr_add(&r, &x, &y); break;
        case BF_SUB: fr_sub(&r, &x, &y); break;
        case BF_MUL: fr_mul(&r, &x, &y); break;
        case BF_DIV: fr_div(&r, &x, &y); break;
        default: fr_from_u64(&r, fr_eq(&x, &y)); break;
        }
        reg_set(vm, o->a, &r);
        break;
    case BR_BINARY_INT_OP:
        x = reg_get(vm, o->b);
        y = reg_get(vm, o->c);
        fr_zero(&r);
        int_op(vm, o->sub_op, o->bit_size, &x, &y, &r);
        if (vm->status == 4) return;
        reg_set(vm, o->a, &r);
        break;
    case BR_JUMP: vm_set_pc(vm, o->location); return;
    case BR_JUMP_IF:
        x = reg_get(vm, o->a);
        if (!fr_is_zero(&x)) { vm_set_pc(vm, o->location); return; }
        break;
    case BR_JUMP_IF_NOT:
        x = reg_get(vm, o->a);
        if (fr_is_zero(&x)) { vm_set_pc(vm, o->location); return; }
        break;
    case BR_RETURN:
        if (vm->n_cs) { vm_set_pc(vm, vm->call_stack[--vm->n_cs] + 1); return; }
        vm_fail(vm, "return opcode hit, but callstack already empty");
        return;
    case BR_CALL:
        if (vm->n_cs == vm->cap_cs) {
            vm->cap_cs = vm->cap_cs ? 2 * vm->cap_cs : 16;
            vm->call_stack = (size_t *)realloc(vm->call_stack, vm->cap_cs * sizeof(size_t));
        }
        vm->call_stack[vm->n_cs++] = vm->pc;
        vm_set_pc(vm, o->location);
        return;
    case BR_CONST: reg_set(vm, o->a, &o->value); break;
    case BR_MOV: x = reg_get(vm, o->b); reg_set(vm, o->a, &x); break;
    case BR_LOAD: { /* a = destination, b = source_pointer */
        const fr_t *s;
        x = reg_get(vm, o->b);
        if (vm->status == 4 || to_usize(vm, &x, &u) || mem_read_slice(vm, u, 1, &s)) return;
        r = *s;
        reg_set(vm, o->a, &r);
        break;
    }
    case BR_STORE: /* a = destination_pointer, b = source */
        x = reg_get(vm, o->a);
        if (vm->status == 4 || to_usize(vm, &x, &u)) return;
        y = reg_get(vm, o->b);
        mem_write_slice(vm, u, &y, 1);
        break;
    case BR_TRAP: vm_fail(vm, "explicit trap hit in brillig"); return;
    case BR_STOP: vm->status = 1; return;
    case BR_BLACK_BOX:
        if (vm_black_box(vm, o)) { vm_fail(vm, vm->msg); return; }
        break;
    case BR_FOREIGN_CALL: { /* lib.rs:194-274 */
        size_t total;
        const fc_result_t *res = fc_result_at(vm, vm->fc_counter, &total);
        if (!res) { /* resolve inputs and wait */
            foreign_call_wait_t *p = &vm->a->pending;
            p->function = strdup(o->function ? o->function : "");
            p->n_inputs = o->n_inputs;
            p->inputs = (fr_t **)calloc(o->n_inputs + 1, sizeof(fr_t *));
            p->input_len = (size_t *)calloc(o->n_inputs + 1, sizeof(size_t));
            for (size_t i = 0; i < o->n_inputs; i++) {
                const reg_or_mem_t *in = &o->inputs[i];
                if (in->kind == ROM_REGISTER) {
                    p->inputs[i] = (fr_t *)malloc(sizeof(fr_t));
                    p->inputs[i][0] = reg_get(vm, in->reg);
                    p->input_len[i] = 1;
                } else {
                    uint64_t start, size = in->size;
                    const fr_t *s;
                    if (reg_usize(vm, in->reg, &start)) return;
                    if (in->kind == ROM_HEAP_VECTOR && reg_usize(vm, in->size, &size)) return;
                    if (mem_read_slice(vm, start, size, &s)) return;
                    p->inputs[i] = (fr_t *)malloc((size + 1) * sizeof(fr_t));
                    memcpy(p->inputs[i], s, size * sizeof(fr_t));
                    p->input_len[i] = size;
                }
            }
            vm->status = 3;
            return;
        }
        int invalid = 0;
        size_t nz = o->n_dests < res->n ? o->n_dests : res->n;
        for (size_t i = 0; i < nz; i++) {
            const reg_or_mem_t *d = &o->dests[i];
            const fc_output_t *out = &res->values[i];
            if (d->kind == ROM_REGISTER) {
                if (out->is_array) { vm_panic(vm, "Function result size does not match brillig bytecode (expected 1 result)"); return; }
                reg_set(vm, d->reg, &out->single);
            } else {
                if (!out->is_array) { vm_panic(vm, "Function result size does not match brillig bytecode size"); return; }
                if (d->kind == ROM_HEAP_ARRAY) {
                    if (out->n != d->size) { invalid = 1; break; }
                } else {
                    fr_t sz;
                    fr_from_u64(&sz, out->n);
                    reg_set(vm, d->size, &sz);
                }
                uint64_t dst;
                if (reg_usize(vm, d->reg, &dst)) return;
                mem_write_slice(vm, dst, out->arr, out->n);
            }
            if (vm->status == 4) return;
        }
        /* lib.rs:262-270: both checks call fail() (which only records status) and execution continues */
        int failed = 0;
        if (o->n_dests != res->n) {
            snprintf(vm->msg, sizeof vm->msg, "%zu output values were provided as a foreign call result for %zu destination slots", res->n, o->n_dests);
            failed = 1;
        }
        if (invalid) { snprintf(vm->msg, sizeof vm->msg, "Function result size does not match brillig bytecode"); failed = 1; }
        vm->fc_counter++;
        if (failed) {
            /* fail() snapshots call_stack + the *current* pc, then increment_program_counter() returns the
             * (Failure) status unless the new pc runs off the end, in which case status becomes Finished. */
            vm->status = 2;
            vm->pc += 1;
            if (vm->pc >= vm->b->n_bytecode) vm->status = 1;
            else vm->pc -= 1; /* report the failing pc */
            return;
        }
        break;
    }
    default: vm_panic(vm, "bad brillig opcode"); return;
    }
    if (vm->status == 4) return;
    vm_set_pc(vm, vm->pc + 1);
}

static int zero_outputs(oracle_acvm_t *a, const brillig_t *b) { /* pwg/brillig.rs:133-150 */
    fr_t z;
    fr_zero(&z);
    for (size_t i = 0; i < b->n_outputs; i++) {
        if (!b->outputs[i].is_array) { if (pwg_insert_value(a, b->outputs[i].w, &z)) return 1; }
        else for (size_t j = 0; j < b->outputs[i].n; j++) if (pwg_insert_value(a, b->outputs[i].arr[j], &z)) return 1;
    }
    return 0;
}

int brillig_solve(oracle_acvm_t *a, const brillig_t *b, size_t acir_index) {
    fr_t pred;
    if (b->has_predicate) { if (pwg_get_value(a, &b->predicate, &pred)) return 1; } /* :28-31 MissingAssignment */
    else fr_one(&pred);
    if (fr_is_zero(&pred)) return zero_outputs(a, b); /* :34-37 */
    vm_t vm;
    memset(&vm, 0, sizeof vm);
    vm.b = b; vm.a = a; vm.acir_index = acir_index;
    int rc = 0;
    /* inputs :46-74 */
    for (size_t i = 0; i < b->n_inputs && !rc; i++) {
        const brillig_input_t *in = &b->inputs[i];
        fr_t v;
        if (!in->is_array) {
            if (pwg_get_value(a, &in->single, &v)) { pwg_fail(a, E_TOO_MANY_UNKNOWNS, 0, 0, NULL); rc = 1; break; }
            reg_set(&vm, vm.n_reg, &v);
        } else {
            uint64_t ptr = vm.n_mem;
            for (size_t j = 0; j < in->n; j++) {
                if (pwg_get_value(a, &in->arr[j], &v)) { pwg_fail(a, E_TOO_MANY_UNKNOWNS, 0, 0, NULL); rc = 1; break; }
                mem_write_slice(&vm, vm.n_mem, &v, 1);
            }
            if (rc) break;
            fr_from_u64(&v, ptr);
            reg_set(&vm, vm.n_reg, &v);
        }
    }
    if (!rc) {
        /* process_opcodes (lib.rs:136-142): process_opcode indexes bytecode[pc] first -> empty bytecode panics */
        do vm_step(&vm); while (vm.status == 0);
        if (vm.status == 1) { /* Finished :95-111 */
            for (size_t i = 0; i < b->n_outputs && !rc; i++) {
                fr_t rv = reg_get(&vm, i);
                if (!b->outputs[i].is_array) rc = pwg_insert_value(a, b->outputs[i].w, &rv);
                else {
                    uint64_t base;
                    if (!fr_try_to_u64(&rv, &base)) { pwg_fail(a, E_PANIC, 0, 0, "register does not fit into u64"); rc = 1; break; }
                    for (size_t j = 0; j < b->outputs[i].n; j++) {
                        if (base + j >= vm.n_mem) { pwg_fail(a, E_PANIC, 0, 0, "index out of bounds: brillig memory"); rc = 1; break; }
                        if (pwg_insert_value(a, b->outputs[i].arr[j], &vm.mem[base + j])) { rc = 1; break; }
                    }
                }
            }
        } else if (vm.status == 2) { /* :113-125 */
            pwg_fail(a, E_BRILLIG_FAILED, 0, 0, vm.msg);
            a->res.n_call_stack = 0;
            for (size_t i = 0; i < vm.n_cs && a->res.n_call_stack < 15; i++) a->res.call_stack[a->res.n_call_stack++] = (uint32_t)vm.call_stack[i];
            a->res.call_stack[a->res.n_call_stack++] = (uint32_t)vm.pc;
            rc = 1;
        } else if (vm.status == 3) rc = 2;
        else { pwg_fail(a, E_PANIC, 0, 0, vm.msg); rc = 1; }
    }
    free(vm.reg); free(vm.mem); free(vm.call_stack);
    return rc;
}
