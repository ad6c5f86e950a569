/*
 * oracle/fr.h -- TEST INFRASTRUCTURE ONLY (CPU oracle). Not part of the shipped product path.
 *
 * CPU restatement of acir_field::FieldElement over BN254-Fr.
 * Follows /root/reference/acir_field/src/generic_ark.rs (line numbers cited per function in fr.c).
 * The underlying arithmetic is ark-ff 0.4.2 / ark-bn254 0.4.0 (Cargo.lock), absent from the
 * reference tree: results are canonical residues mod p, fully determined by p, so any correct
 * implementation is bit-exact. Pinned by tests/test_oracle_fr.py against the reference's own
 * vectors (generic_ark.rs:423-438, acvm_js/test/shared/foreign_call.ts) and Python big-ints.
 */
#ifndef ORACLE_FR_H
#define ORACLE_FR_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Montgomery form, R = 2^256, little-endian 64-bit limbs (same in-memory shape as ark-ff Fp256). */
typedef struct {
    uint64_t l[4];
} fr_t;

extern const uint64_t FR_MODULUS[4];

void fr_zero(fr_t *r);
void fr_one(fr_t *r);
void fr_from_u64(fr_t *r, uint64_t v);
/* generic_ark.rs:281-283 from_be_bytes_reduce (any length, reduced mod p) */
void fr_from_be_bytes_reduce(fr_t *r, const uint8_t *bytes, size_t len);
/* generic_ark.rs:269-277 to_be_bytes (canonical 32-byte big-endian) */
void fr_to_be_bytes(const fr_t *a, uint8_t out[32]);
/* canonical little-endian limbs (non-Montgomery) */
void fr_to_canonical(const fr_t *a, uint64_t out[4]);
void fr_from_canonical(fr_t *r, const uint64_t in[4]); /* in < p required */

void fr_add(fr_t *r, const fr_t *a, const fr_t *b);
void fr_sub(fr_t *r, const fr_t *a, const fr_t *b);
void fr_neg(fr_t *r, const fr_t *a);
void fr_mul(fr_t *r, const fr_t *a, const fr_t *b);
/* generic_ark.rs:242-245: inverse(0) == 0 */
void fr_inverse(fr_t *r, const fr_t *a);
/* generic_ark.rs:375-381: a / b = a * inverse(b) */
void fr_div(fr_t *r, const fr_t *a, const fr_t *b);
int fr_is_zero(const fr_t *a);
int fr_is_one(const fr_t *a);
int fr_eq(const fr_t *a, const fr_t *b);
/* canonical integer order (derived Ord on the ark field) : -1,0,1 */
int fr_cmp(const fr_t *a, const fr_t *b);
/* generic_ark.rs:214-221 */
uint32_t fr_num_bits(const fr_t *a);
/* generic_ark.rs:227-230: low 128 bits, silently truncating */
void fr_to_u128(const fr_t *a, uint64_t *lo, uint64_t *hi);
/* generic_ark.rs:236-238: returns 1 and sets *v if num_bits <= 64 */
int fr_try_to_u64(const fr_t *a, uint64_t *v);
/* generic_ark.rs:305-317: low ceil(num_bits/8) bytes, least-significant first.
 * Returns the byte count, or -1 where the reference would panic (slice past 32 bytes). */
int fr_fetch_nearest_bytes(const fr_t *a, uint32_t num_bits, uint8_t out[32]);
/* generic_ark.rs:328-355 + mask_vector_le :446-473 */
void fr_and_xor(fr_t *r, const fr_t *a, const fr_t *b, uint32_t num_bits, int is_xor);
/* 64 lowercase hex chars, no 0x (serde form, generic_ark.rs:114-121,257-262). out needs 65 bytes. */
void fr_to_hex(const fr_t *a, char out[65]);
/* generic_ark.rs:263-267 from_hex: optional 0x, decode, reduce. returns 0 on success */
int fr_from_hex(fr_t *r, const char *s, size_t len);

#ifdef __cplusplus
}
#endif
#endif
