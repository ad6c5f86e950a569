"""The curve arithmetic of the ECDSA kernels (acvm_amd/csrc/secp_device.hpp) run ON THE DEVICE against Python integers, through the
acvm_debug_secp probes: the same cases tests/test_secp_device_on_host.py runs through the host build of that header. The reference's
arithmetic is k256 0.11.6 / p256 0.11.1 behind blackbox_solver/src/lib.rs:66-210; integers modulo the two primes are the spec."""
import random

import pytest

from test_secp_device_on_host import CURVES, affine, ec_add, ec_mul, edge_values

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("c", [0, 1])
def test_field_arithmetic_on_device(c):
    import acvm_amd
    p = CURVES[c]["p"]
    rng = random.Random(4000 + c)
    ev = edge_values(p)
    pairs = [(a, b) for a in ev for b in ev] + [(rng.randrange(p), rng.randrange(p)) for _ in range(4000)]
    for _ in range(2000):  # long runs of ones / zeros drive the carries of the reductions
        a = rng.choice([0, 2**256 - 1]) ^ (((1 << rng.randrange(1, 256)) - 1) << rng.randrange(0, 256))
        b = rng.choice([0, 2**256 - 1]) ^ (((1 << rng.randrange(1, 256)) - 1) << rng.randrange(0, 256))
        pairs.append((a % 2**256 % p, b % 2**256 % p))
    for what, name, f in ((0, "mul", lambda a, b: a * b % p), (2, "add", lambda a, b: (a + b) % p), (3, "sub", lambda a, b: (a - b) % p)):
        got = acvm_amd.debug_secp(c, what, pairs)
        bad = [(name, hex(a), hex(b), hex(g[0]), hex(f(a, b))) for (a, b), g in zip(pairs, got) if g[0] != f(a, b)]
        assert not bad, bad[:3]
    for what, name, f in ((9, "(a+b)^2", lambda a, b: (a + b) ** 2 % p), (10, "a-4b", lambda a, b: (a - 4 * b) % p)):
        got = acvm_amd.debug_secp(c, what, pairs)
        bad = [(name, hex(a), hex(b), hex(g[0]), hex(f(a, b))) for (a, b), g in zip(pairs, got) if g[0] != f(a, b)]
        assert not bad, bad[:3]
    ones = [(a,) for a, _ in pairs]
    for what, e in ((8, 4), (11, 32)):
        got = acvm_amd.debug_secp(c, what, ones)
        bad = [(f"a^{e}", hex(a), hex(g[0]), hex(pow(a, e, p))) for (a,), g in zip(ones, got) if g[0] != pow(a, e, p)]
        assert not bad, bad[:3]
    got = acvm_amd.debug_secp(c, 1, ones)
    bad = [("sqr", hex(a), hex(g[0])) for (a,), g in zip(ones, got) if g[0] != a * a % p]
    assert not bad, bad[:3]
    got = acvm_amd.debug_secp(c, 4, ones[:600])
    bad = [("inv", hex(a), hex(g[0])) for (a,), g in zip(ones[:600], got) if g[0] != (pow(a, -1, p) if a else 0)]
    assert not bad, bad[:3]
    sq = [(a * a % p,) for a, _ in pairs[200:500]]
    got = acvm_amd.debug_secp(c, 5, sq)
    bad = [("sqrt", hex(a), hex(g[0])) for (a,), g in zip(sq, got) if g[0] != pow(a, (p + 1) // 4, p)]
    assert not bad, bad[:3]


@pytest.mark.parametrize("c", [0, 1])
def test_point_formulas_on_device(c):
    import acvm_amd
    cv = CURVES[c]
    p, n = cv["p"], cv["n"]
    rng = random.Random(5000 + c)
    dbl, add, want_d, want_a = [], [], [], []
    for it in range(300):
        P, Q = ec_mul(cv, rng.randrange(1, n), cv["g"]), ec_mul(cv, rng.randrange(1, n), cv["g"])
        z = rng.randrange(1, p)
        X, Y, Z = P[0] * z * z % p, P[1] * z * z * z % p, z
        dbl.append((X, Y, Z)); want_d.append(ec_add(cv, P, P))
        if it % 10 == 0: Q = P                      # the addition that is a doubling
        if it % 10 == 1: Q = (P[0], (p - P[1]) % p)  # ... and the one that cancels
        add.append((X, Y, Z, Q[0], Q[1])); want_a.append(ec_add(cv, P, Q))
    add.append((1, 1, 0, cv["g"][0], cv["g"][1])); want_a.append(cv["g"])  # identity + G
    dbl.append((1, 1, 0)); want_d.append(None)
    for what, items, want in ((6, dbl, want_d), (7, add, want_a)):
        got = acvm_amd.debug_secp(c, what, items)
        for it, g, w in zip(items, got, want):
            assert affine(cv, *g) == w, (what, it, g)
