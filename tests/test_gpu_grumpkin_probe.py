"""Intermediate results of the Grumpkin device routines (acvm_debug_grumpkin probes) against the CPU oracle and against
plain affine arithmetic in Python: device table points, hash_single, the hash-ladder compress, fixed-base multiples, the
endomorphism constant and the complete addition formulas on the lazy 29-bit working form."""
import ctypes as C
import random

import pytest

import acvm_amd
from acvm_amd.acir import P

pytestmark = pytest.mark.gpu
BETA = 0xb3c4d79d41a917585bfc41088d8daaa78b17ea66b99c90dd


def xy(b):
    return (int.from_bytes(b.raw[:32], "big"), int.from_bytes(b.raw[32:], "big"))


def aff_add(p, q):
    if p is None:
        return q
    if q is None:
        return p
    if p[0] == q[0]:
        if (p[1] + q[1]) % P == 0:
            return None
        lam = 3 * p[0] * p[0] * pow(2 * p[1], -1, P) % P
    else:
        lam = (q[1] - p[1]) * pow(q[0] - p[0], -1, P) % P
    x = (lam * lam - p[0] - q[0]) % P
    return (x, (lam * (p[0] - x) - p[1]) % P)


def test_device_tables_match_host_tables():
    for prm in [0, 511, 512 * 29 + 3, (1 << 24) | 0, (1 << 24) | (32 * 255 * 3 + 31 * 255 + 254), (2 << 24) | 44, (3 << 24) | 2]:
        assert acvm_amd.debug_grumpkin(0, prm) == acvm_amd.debug_grumpkin(4, prm), hex(prm)


def test_hash_single_compress_and_fixed_base_against_oracle(oracle):
    r = random.Random(5)
    out = C.create_string_buffer(64)
    for v in [0, 1, 2, 511, 512, P - 1, r.randrange(P), r.randrange(P)]:
        for par in (0, 1):
            oracle.lib().oracle_pedersen_hash_single(v.to_bytes(32, "big"), par, out)
            assert xy(out) == acvm_amd.debug_grumpkin(1, par, [v]), (hex(v), par)
    for vs in [[1], [0, 1, 2], [P - 1, 5, 6], [r.randrange(P) for _ in range(3)]]:
        o32 = C.create_string_buffer(32)
        oracle.lib().oracle_pedersen_compress(b"".join(v.to_bytes(32, "big") for v in vs), len(vs), o32)
        assert int.from_bytes(o32.raw, "big") == acvm_amd.debug_grumpkin(2, 0, vs)[0]
    for k in [1, 2, 255, 256, r.randrange(1 << 254)]:
        oracle.lib().oracle_grumpkin_mul_g(k.to_bytes(32, "big"), out)
        assert xy(out) == acvm_amd.debug_grumpkin(3, 0, [k])


def test_point_arithmetic_against_affine_formulas():
    tbl = lambda i: acvm_amd.debug_grumpkin(0, i)  # noqa: E731
    for v in [0, 12345678901234567890123]:
        for par in (0, 1):
            acc = None
            for i in range(15):
                acc = aff_add(acc, tbl((par * 15 + i) * 512 + ((v >> (18 * i)) & 511)))
            assert acvm_amd.debug_grumpkin(5, par, [v]) == acc
    for v in [1, 5, P - 1, 0x1234567890abcdef << 100]:
        g = acvm_amd.debug_grumpkin(6, 0, [v])
        assert g[0] == v * BETA % P and g[1] == BETA
    p0, p1 = tbl(0), tbl(512)
    assert acvm_amd.debug_grumpkin(7, 0) == aff_add(p0, aff_add(p1, p0))


def test_var_base_mul_glv_against_affine_double_and_add():
    """e * P as SchnorrVerify computes it (GLV split e = k1 + k2 lambda, joint 4-bit windows over the lane's table; probe 8)
    against the device's plain double-and-add (probe 9) and against affine arithmetic in Python, on scalars that exercise the
    split: 0, 1, q - 1, multiples of lambda (k1 = 0), negative halves, 2^127 boundaries, random."""
    Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
    LAM = 0x59e26bcea0d48bacd4f263f1acdb5c4f5763473177fffffe

    def mul(k, pt):
        acc = None
        while k:
            if k & 1:
                acc = aff_add(acc, pt)
            pt = aff_add(pt, pt)
            k >>= 1
        return acc

    g = acvm_amd.debug_grumpkin(3, 0, [1])                 # the generator through the fixed-base path
    assert mul(LAM, g) == (g[0] * BETA % P, g[1])          # the endomorphism pair the split relies on
    pts = [g, mul(0xC0FFEE1234567, g)]
    r = random.Random(11)
    scalars = [0, 1, 2, 15, 16, Q - 1, Q - 2, LAM, LAM + 1, LAM - 1, Q - LAM, 2 * LAM % Q, 7 * LAM % Q, (Q - 3 * LAM) % Q, 1 << 127, (1 << 127) - 1,
               (1 << 128) + 5, 1 << 253, Q // 2, Q // 3] + [r.randrange(Q) for _ in range(12)]
    for pt in pts:
        for k in scalars:
            want = mul(k, pt) or (0, 0)                    # infinity is exported as (0, 0)
            assert acvm_amd.debug_grumpkin(8, 0, [k, pt[0], pt[1]]) == want, hex(k)
        for k in scalars[:8] + scalars[-2:]:
            assert acvm_amd.debug_grumpkin(9, 0, [k, pt[0], pt[1]]) == (mul(k, pt) or (0, 0)), hex(k)
