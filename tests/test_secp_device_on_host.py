"""The curve arithmetic behind the ECDSA opcodes (acvm_amd/csrc/secp_device.hpp is __host__ __device__) executed on the host against
Python integers: field products / squares with the prime-shaped reductions at the edges of their carries, add / sub, both inversions,
the square-root chains, the point formulas, entries of the generator table, and whole verifications (valid, tampered, high-S, the
panics of blackbox_solver/src/lib.rs:101-210). No GPU is needed: hipcc builds the host side of tools/secp_device_host_test.hip."""
import os
import random
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CURVES = {
    0: dict(p=2**256 - 2**32 - 977, n=0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141, a=0, b=7,
            g=(0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798, 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8)),
    1: dict(p=2**256 - 2**224 + 2**192 + 2**96 - 1, n=0xFFFFFFFF00000000FFFFFFFFFFFFFFFFBCE6FAADA7179E84F3B9CAC2FC632551, a=-3,
            b=0x5AC635D8AA3A93E7B3EBBD55769886BC651D06B0CC53B0F63BCE3C3E27D2604B,
            g=(0x6B17D1F2E12C4247F8BCE6E563A440F277037D812DEB33A0F4A13945D898C296, 0x4FE342E2FE1A7F9B8EE7EB4A7C0F9E162BCE33576B315ECECBB6406837BF51F5)),
}


def ec_add(cv, P, Q):
    p = cv["p"]
    if P is None: return Q
    if Q is None: return P
    if P[0] == Q[0]:
        if (P[1] + Q[1]) % p == 0: return None
        lam = (3 * P[0] * P[0] + cv["a"]) * pow(2 * P[1], -1, p) % p
    else:
        lam = (Q[1] - P[1]) * pow(Q[0] - P[0], -1, p) % p
    x = (lam * lam - P[0] - Q[0]) % p
    return x, (lam * (P[0] - x) - P[1]) % p


def ec_mul(cv, k, P):
    R = None
    while k:
        if k & 1: R = ec_add(cv, R, P)
        P = ec_add(cv, P, P)
        k >>= 1
    return R


def affine(cv, X, Y, Z):
    p = cv["p"]
    if Z % p == 0: return None
    zi = pow(Z, -1, p)
    return X * zi * zi % p, Y * zi * zi * zi % p


def verify_model(cv, r, s, x, y_odd, n_msg, z):
    """(result, panic) in the order of the device routine / the reference's unwraps"""
    p, n = cv["p"], cv["n"]
    if r == 0 or s == 0 or r >= n or s >= n: return 0, 1
    if x >= p: return 0, 2
    rhs = (x * x * x + cv["a"] * x + cv["b"]) % p
    y = pow(rhs, (p + 1) // 4, p)
    if y * y % p != rhs: return 0, 2
    if (y & 1) != (y_odd & 1): y = (p - y) % p
    if n_msg != 32: return 0, 3
    if z >= n: return 0, 4
    if s > n // 2: return 0, 0
    si = pow(s, -1, n)
    R = ec_add(cv, ec_mul(cv, z * si % n, cv["g"]), ec_mul(cv, r * si % n, (x, y)))
    if R is None: return 0, 5
    if R[0] >= n: return 0, 6
    return (1 if R[0] == r else 0), 0


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    out = str(tmp_path_factory.mktemp("secp") / "secp_host_test")
    # (-DSECP_GWIN_BITS=8: the same routines over an 8-bit generator table; the device's 16-bit one would take minutes to build on the host)
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O2", "-DSECP_GWIN_BITS=8", "-DSECP_CHECK", os.path.join(ROOT, "tools", "secp_device_host_test.hip"), "-o", out], check=True, timeout=900)
    return out


def ask(exe, lines):
    out = subprocess.run([exe], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout.split("\n")[:len(lines)]


def edge_values(m):
    vals = [0, 1, 2, m - 1, m - 2, 2**32 - 1, 2**32, 2**64 - 1, 2**128, 2**255 % m, (2**256 - 1) % m, m >> 1, 977, 2**224 % m, (2**224 - 2**192) % m]
    return [v % m for v in vals]


@pytest.mark.parametrize("c", [0, 1])
def test_field_arithmetic(exe, c):
    cv = CURVES[c]
    p, n = cv["p"], cv["n"]
    rng = random.Random(1000 + c)
    ev = edge_values(p)
    pairs = [(a, b) for a in ev for b in ev] + [(rng.randrange(p), rng.randrange(p)) for _ in range(3000)]
    # operands with long runs of ones / zeros drive the carries of the reductions
    for _ in range(1000):
        a = rng.choice([0, 2**256 - 1]) ^ (((1 << rng.randrange(1, 256)) - 1) << rng.randrange(0, 256))
        b = rng.choice([0, 2**256 - 1]) ^ (((1 << rng.randrange(1, 256)) - 1) << rng.randrange(0, 256))
        pairs.append((a % 2**256 % p, b % 2**256 % p))
    lines, want = [], []
    for a, b in pairs:
        lines += [f"{c} mul {a:x} {b:x}", f"{c} sqr {a:x}", f"{c} add {a:x} {b:x}", f"{c} sub {a:x} {b:x}"]
        want += [a * b % p, a * a % p, (a + b) % p, (a - b) % p]
    for a, _ in pairs[:400]:
        lines += [f"{c} inv {a:x}", f"{c} ninv {a % n:x}", f"{c} nmul {a % n:x} {(a * 7 + 3) % n:x}"]
        want += [pow(a, -1, p) if a else 0, pow(a % n, -1, n) if a % n else 0, (a % n) * ((a * 7 + 3) % n) % n]
    for a, _ in pairs[225:325]:
        sq = a * a % p
        lines.append(f"{c} sqrt {sq:x}")
        want.append(pow(sq, (p + 1) // 4, p))
    got = ask(exe, lines)
    bad = [(l, g, f"{w:064x}") for l, g, w in zip(lines, got, want) if g.strip() != f"{w:064x}"]
    assert not bad, bad[:5]


@pytest.mark.parametrize("c", [0, 1])
def test_point_formulas_and_generator_table(exe, c):
    cv = CURVES[c]
    p, n = cv["p"], cv["n"]
    rng = random.Random(2000 + c)
    lines, want = [], []

    def jac(P):
        z = rng.randrange(1, p)
        return P[0] * z * z % p, P[1] * z * z * z % p, z

    for it in range(60):
        k1, k2 = rng.randrange(1, n), rng.randrange(1, n)
        P, Q = ec_mul(cv, k1, cv["g"]), ec_mul(cv, k2, cv["g"])
        X, Y, Z = jac(P)
        lines.append(f"{c} dbl {X:x} {Y:x} {Z:x}"); want.append(ec_add(cv, P, P))
        if it % 10 == 0: Q = P                      # the addition that is a doubling
        if it % 10 == 1: Q = (P[0], (p - P[1]) % p)  # ... and the one that cancels
        lines.append(f"{c} addaff {X:x} {Y:x} {Z:x} {Q[0]:x} {Q[1]:x}"); want.append(ec_add(cv, P, Q))
    lines.append(f"{c} addaff 1 1 0 {cv['g'][0]:x} {cv['g'][1]:x}"); want.append(cv["g"])  # identity + G
    lines.append(f"{c} dbl 1 1 0"); want.append(None)
    got = ask(exe, lines)
    for l, g, w in zip(lines, got, want):
        X, Y, Z = (int(t, 16) for t in g.split())
        assert affine(cv, X, Y, Z) == w, l
    lines, want = [], []
    for j, d in [(0, 1), (0, 255), (1, 1), (7, 130), (31, 255), (31, 1), (16, 77)]:
        lines.append(f"{c} gtab {j} {d}"); want.append(ec_mul(cv, d << (8 * j), cv["g"]))
    got = ask(exe, lines)
    for l, g, w in zip(lines, got, want):
        assert tuple(int(t, 16) for t in g.split()) == w, l


def test_secp256k1_endomorphism_split(exe):
    """k = k1 + k2 lambda (mod n) with both parts below 2^128 in magnitude (secp_device.hpp secp256k1_split_lambda)"""
    n = CURVES[0]["n"]
    lam = 0x5363AD4CC05C30E0A5261C028812645A122E22EA20816678DF02967C1B23BD72
    rng = random.Random(77)
    ks = [0, 1, 2, n - 1, n - 2, n // 2, n // 2 + 1, lam, n - lam, 2**128, 2**128 - 1, 2**255, (2**256 - 1) % n] + [rng.randrange(n) for _ in range(3000)]
    got = ask(exe, [f"0 split {k:x}" for k in ks])
    for k, g in zip(ks, got):
        k1, s1, k2, s2 = g.split()
        k1, k2 = int(k1, 16), int(k2, 16)
        assert k1 < 2**128 and k2 < 2**128, hex(k)
        assert ((-k1 if s1 == "1" else k1) + (-k2 if s2 == "1" else k2) * lam) % n == k, hex(k)


@pytest.mark.parametrize("c", [0, 1])
def test_verification(exe, c):
    cv = CURVES[c]
    p, n = cv["p"], cv["n"]
    rng = random.Random(3000 + c)
    cases = []

    def sign(d, z, k):
        R = ec_mul(cv, k, cv["g"])
        r = R[0] % n
        s = pow(k, -1, n) * (z + r * d) % n
        return r, s

    for it in range(40):
        d, z, k = rng.randrange(1, n), rng.randrange(n), rng.randrange(1, n)
        if it == 0: z = 0
        if it == 1: z = n - 1
        Q = ec_mul(cv, d, cv["g"])
        r, s = sign(d, z, k)
        low = min(s, n - s)
        cases.append((r, low, Q[0], Q[1] & 1, 32, z))             # valid, low-S
        if it % 4 == 0: cases.append((r, n - low, Q[0], Q[1] & 1, 32, z))   # high-S: false
        if it % 4 == 1: cases.append((r, low, Q[0], (Q[1] & 1) ^ 1, 32, z)) # -Q: false
        if it % 4 == 2: cases.append((r, low, Q[0], Q[1] & 1, 32, (z + 1) % n))
        if it % 4 == 3: cases.append(((r + 1) % n or 1, low, Q[0], Q[1] & 1, 32, z))
    Q = ec_mul(cv, 5, cv["g"])
    cases += [(0, 1, Q[0], 0, 32, 5), (1, 0, Q[0], 0, 32, 5), (n, 1, Q[0], 0, 32, 5), (1, n, Q[0], 0, 32, 5),   # signature scalars out of range
              (1, 1, p, 0, 32, 5), (1, 1, 2**256 - 1, 0, 32, 5),                                              # x >= p
              (1, 1, Q[0], 0, 31, 5), (1, 1, Q[0], 0, 33, 5), (1, 1, Q[0], 0, 32, n), (1, 1, Q[0], 0, 32, 2**256 - 1)]
    x = 1
    while pow((x ** 3 + cv["a"] * x + cv["b"]) % p, (p - 1) // 2, p) == 1: x += 1
    cases.append((1, 1, x, 0, 32, 5))                                                                          # x not on the curve
    # u1 G + u2 Q = identity: Q = G, u1 = -u2  <=>  z = -r (mod n)
    cases.append((7, 3, cv["g"][0], cv["g"][1] & 1, 32, n - 7))
    # small scalars, u2 with a top carry in the signed recoding (r / s close to n)
    cases.append((n - 1, 1, Q[0], Q[1] & 1, 32, n - 1))
    cases.append((1, 1, Q[0], Q[1] & 1, 32, 0))
    # u2 = r / s with special shapes (secp256k1: the halves of the endomorphism split): u2 = 1, 2^128, lambda, n - 1 for s = r / u2
    for u2 in (1, 2, 2**128, 2**128 - 1, 0x5363AD4CC05C30E0A5261C028812645A122E22EA20816678DF02967C1B23BD72, n - 1, n - 2**128):
        d, k = rng.randrange(1, n), rng.randrange(1, n)
        Qd = ec_mul(cv, d, cv["g"])
        r = ec_mul(cv, k, cv["g"])[0] % n
        s = r * pow(u2, -1, n) % n
        z = (s * k - r * d) % n   # k = (z + r d) / s
        if 0 < s <= n // 2:
            cases.append((r, s, Qd[0], Qd[1] & 1, 32, z))
        else:
            cases.append((r, n - s, Qd[0], Qd[1] & 1, 32, z))   # high-S counterpart: rejected or, mirrored, another valid-looking input
    lines = [f"{c} verify {r:x} {s:x} {x:x} {yo} {nm} {z:x}" for r, s, x, yo, nm, z in cases]
    got = ask(exe, lines)
    seen = set()
    for l, g, cs in zip(lines, got, cases):
        want = verify_model(cv, *cs)
        assert tuple(int(t) for t in g.split()) == want, (l, g, want)
        seen.add(want)
    assert {(1, 0), (0, 0), (0, 1), (0, 2), (0, 3), (0, 4), (0, 5)} <= seen
    # the same cases with the WHOLE of public_key_y given (the decompression shortcut of secp_verify): the true root of that parity, an
    # off-curve value of that parity, an unreduced one (>= p) of that parity -- the outcome only ever depends on x and the parity
    lines, wants = [], []
    for r, s, x, yo, nm, z in cases:
        want = verify_model(cv, r, s, x, yo, nm, z)
        ys = [((2**256 - 2) | (yo & 1)), ((12345678 << 1) | (yo & 1))]
        if x < p:
            rhs = (x * x * x + cv["a"] * x + cv["b"]) % p
            y = pow(rhs, (p + 1) // 4, p)
            if y * y % p == rhs:
                ys.append(y if (y & 1) == (yo & 1) else p - y)
        for y in ys:
            lines.append(f"{c} verifyy {r:x} {s:x} {x:x} {y:x} {nm} {z:x}")
            wants.append(want)
    for l, g, want in zip(lines, ask(exe, lines), wants):
        assert tuple(int(t) for t in g.split()) == want, (l, g, want)
