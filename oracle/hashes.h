/* oracle/hashes.h -- TEST INFRASTRUCTURE ONLY (CPU oracle). See hashes.c for provenance. */
#ifndef ORACLE_HASHES_H
#define ORACLE_HASHES_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
void oracle_sha256(const uint8_t *msg, size_t len, uint8_t out[32]);
void oracle_keccak256(const uint8_t *msg, size_t len, uint8_t out[32]);
void oracle_keccak_f1600(uint64_t s[25]);
void oracle_blake2s(const uint8_t *msg, size_t len, uint8_t out[32]);
#ifdef __cplusplus
}
#endif
#endif
