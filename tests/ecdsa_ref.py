"""Independent big-integer model of ECDSA over secp256k1 / secp256r1 (SEC 1 v2 4.1.3 / 4.1.4) with the call-site rules of
blackbox_solver/src/lib.rs:101-210, used to pin the C oracle (tests/test_oracle_ecdsa.py) and to make signatures for the
GPU parity tests. Test infrastructure only."""
CURVES = {
    0: dict(p=0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEFFFFFC2F, n=0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141, a=0, b=7,
            g=(0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798, 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8)),
    1: dict(p=0xFFFFFFFF00000001000000000000000000000000FFFFFFFFFFFFFFFFFFFFFFFF, n=0xFFFFFFFF00000000FFFFFFFFFFFFFFFFBCE6FAADA7179E84F3B9CAC2FC632551, a=-3,
            b=0x5AC635D8AA3A93E7B3EBBD55769886BC651D06B0CC53B0F63BCE3C3E27D2604B,
            g=(0x6B17D1F2E12C4247F8BCE6E563A440F277037D812DEB33A0F4A13945D898C296, 0x4FE342E2FE1A7F9B8EE7EB4A7C0F9E162BCE33576B315ECECBB6406837BF51F5)),
}


def add(c, P, Q):
    p = c["p"]
    if P is None:
        return Q
    if Q is None:
        return P
    if P[0] == Q[0]:
        if (P[1] + Q[1]) % p == 0:
            return None
        lam = (3 * P[0] * P[0] + c["a"]) * pow(2 * P[1], -1, p) % p
    else:
        lam = (Q[1] - P[1]) * pow(Q[0] - P[0], -1, p) % p
    x = (lam * lam - P[0] - Q[0]) % p
    return x, (lam * (P[0] - x) - P[1]) % p


def mul(c, k, P):
    R = None
    for bit in bin(k)[2:] if k else "":
        R = add(c, R, R)
        if bit == "1":
            R = add(c, R, P)
    return R


def sign(curve, sk, k, z, low_s=True):
    c = CURVES[curve]
    n = c["n"]
    R = mul(c, k, c["g"])
    r = R[0] % n
    s = pow(k, -1, n) * (z + r * sk) % n
    if low_s and s > n // 2:
        s = n - s
    return r, s


def public_key(curve, sk):
    return mul(CURVES[curve], sk, CURVES[curve]["g"])


def verify(curve, z_bytes, pkx, pky, sig):
    """1 / 0, or a negative panic code numbered like oracle/ecdsa.c."""
    c = CURVES[curve]
    n, p = c["n"], c["p"]
    r, s = int.from_bytes(sig[:32], "big"), int.from_bytes(sig[32:], "big")
    if not (0 < r < n and 0 < s < n):
        return -1
    x = int.from_bytes(pkx, "big")
    if x >= p:
        return -2
    rhs = (x ** 3 + c["a"] * x + c["b"]) % p
    y = pow(rhs, (p + 1) // 4, p)
    if y * y % p != rhs:
        return -2
    if (y & 1) != (pky[31] & 1):
        y = p - y
    if len(z_bytes) != 32:
        return -3
    z = int.from_bytes(z_bytes, "big")
    if z >= n:
        return -4
    if s > n // 2:
        return 0
    si = pow(s, -1, n)
    R = add(c, mul(c, z * si % n, c["g"]), mul(c, r * si % n, (x, y)))
    if R is None:
        return -5
    if R[0] >= n:
        return -6
    return int(R[0] == r)
