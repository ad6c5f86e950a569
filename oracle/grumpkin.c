/* placeholder: replaced by the barretenberg restatement */
#include "pwg.h"
#include <stdio.h>
static int u1(void *c, const fr_t *x, const fr_t *y, const uint8_t *s, size_t sl, const uint8_t *m, size_t ml, int *ok, char *err, size_t n) { (void)c;(void)x;(void)y;(void)s;(void)sl;(void)m;(void)ml;(void)ok;(void)err;(void)n; return 2; }
static int u2(void *c, const fr_t *in, size_t k, uint32_t ds, fr_t *x, fr_t *y, char *err, size_t n) { (void)c;(void)in;(void)k;(void)ds;(void)x;(void)y;(void)err;(void)n; return 2; }
static int u3(void *c, const fr_t *lo, const fr_t *hi, fr_t *x, fr_t *y, char *err, size_t n) { (void)c;(void)lo;(void)hi;(void)x;(void)y;(void)err;(void)n; return 2; }
const backend_t ORACLE_BARRETENBERG_BACKEND = {0, u1, u2, u3};
