/*
 * oracle/backends.c -- TEST INFRASTRUCTURE ONLY (CPU oracle). Not part of the shipped product path.
 * The two fakes the reference's own tests use for BlackBoxFunctionSolver, plus the selector.
 *   StubbedBackend       /root/reference/acvm/tests/solver.rs:20-46   (panics when any function is hit)
 *   DummyBlackBoxSolver  /root/reference/brillig_vm/src/lib.rs:392-420 (true, (2,3), (4,5))
 */
#include "pwg.h"
#include <stdio.h>

extern const backend_t ORACLE_BARRETENBERG_BACKEND; /* grumpkin.c */

static int stub_schnorr(void *c, const fr_t *x, const fr_t *y, const uint8_t *s, size_t sl, const uint8_t *m, size_t ml, int *ok, char *err, size_t n) {
    (void)c; (void)x; (void)y; (void)s; (void)sl; (void)m; (void)ml; (void)ok;
    snprintf(err, n, "Path not trodden by this test");
    return 3;
}
static int stub_pedersen(void *c, const fr_t *in, size_t k, uint32_t ds, fr_t *x, fr_t *y, char *err, size_t n) {
    (void)c; (void)in; (void)k; (void)ds; (void)x; (void)y;
    snprintf(err, n, "Path not trodden by this test");
    return 3;
}
static int stub_fixed(void *c, const fr_t *lo, const fr_t *hi, fr_t *x, fr_t *y, char *err, size_t n) {
    (void)c; (void)lo; (void)hi; (void)x; (void)y;
    snprintf(err, n, "Path not trodden by this test");
    return 3;
}
static int dummy_schnorr(void *c, const fr_t *x, const fr_t *y, const uint8_t *s, size_t sl, const uint8_t *m, size_t ml, int *ok, char *err, size_t n) {
    (void)c; (void)x; (void)y; (void)s; (void)sl; (void)m; (void)ml; (void)err; (void)n;
    *ok = 1;
    return 0;
}
static int dummy_pedersen(void *c, const fr_t *in, size_t k, uint32_t ds, fr_t *x, fr_t *y, char *err, size_t n) {
    (void)c; (void)in; (void)k; (void)ds; (void)err; (void)n;
    fr_from_u64(x, 2);
    fr_from_u64(y, 3);
    return 0;
}
static int dummy_fixed(void *c, const fr_t *lo, const fr_t *hi, fr_t *x, fr_t *y, char *err, size_t n) {
    (void)c; (void)lo; (void)hi; (void)err; (void)n;
    fr_from_u64(x, 4);
    fr_from_u64(y, 5);
    return 0;
}
static const backend_t STUBBED = {0, stub_schnorr, stub_pedersen, stub_fixed};
static const backend_t DUMMY = {0, dummy_schnorr, dummy_pedersen, dummy_fixed};

const backend_t *oracle_backend(int which) {
    if (which == 1) return &STUBBED;
    if (which == 2) return &DUMMY;
    return &ORACLE_BARRETENBERG_BACKEND;
}
