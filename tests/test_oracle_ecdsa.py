"""Pins oracle/ecdsa.c: the two vectors of blackbox_solver/src/lib.rs:216-284 and an independent big-integer model
(tests/ecdsa_ref.py) over valid, tampered, high-S and panicking inputs."""
import random

import pytest

from ecdsa_ref import CURVES, public_key, sign, verify

K1 = dict(z="3a73f4123a5cd2121f21cd7e8d358835476949d035d9c2da6806b4633ac8c1e2", x="a0434d9e47f3c86235477c7b1ae6ae5d3442d49b1943c2b752a68e2a47e247c7",
          y="893aba425419bc27a3b6c7e693a24c696f794c2ed877a1593cbee53b037368d7",
          sig="e5081c80ab427dc370346f4a0e31aa2bad8d9798c38061db9ae55a4e8df454fd28119894344e71b78770cc931d61f480ecbb0b89d6eb69690161e49a715fcd55")
R1 = dict(z="54705ba3baafdbdfba8c5f9a70f7a89bee98d906b53e31074da7baecdc0da9ad", x="550f471003f3df97c3df506ac797f6721fb1a1fb7b8f6f83d224498a65c88e24",
          y="136093d7012e509a73715cbd0b00a3cc0ff4b5c01b3ffa196ab1fb327036b8e6",
          sig="2c70a8d084b62bfc5ce03641caf9f72ad4da8c81bfe6ec9487bb5e1bef62a13218ad9ee29eaf351fdc50f1520c425e9b908a07278b43b0ec7b872778c14e0784")


def ov(oracle, curve, z, x, y, sig):
    return oracle.lib().oracle_ecdsa_verify(curve, z, len(z), x, y, sig)


def test_reference_vectors(oracle):
    for curve, v in ((0, K1), (1, R1)):
        args = [bytes.fromhex(v[k]) for k in ("z", "x", "y", "sig")]
        assert ov(oracle, curve, *args) == 1
        assert verify(curve, *args) == 1
        bad = bytearray(args[3])
        bad[40] ^= 1
        assert ov(oracle, curve, args[0], args[1], args[2], bytes(bad)) == verify(curve, args[0], args[1], args[2], bytes(bad))


@pytest.mark.parametrize("curve", [0, 1])
def test_against_big_integer_model(oracle, curve):
    r = random.Random(100 + curve)
    c = CURVES[curve]
    be = lambda v: int(v).to_bytes(32, "big")  # noqa: E731
    cases = []
    for i in range(12):
        sk, k, z = r.randrange(1, c["n"]), r.randrange(1, c["n"]), r.randrange(c["n"])
        Q = public_key(curve, sk)
        rr, ss = sign(curve, sk, k, z, low_s=(i % 4 != 3))
        cases.append((be(z), be(Q[0]), be(Q[1]), be(rr) + be(ss)))            # valid (or high-S every 4th)
        cases.append((be(z), be(Q[0]), be(Q[1] ^ 2), be(rr) + be(ss)))        # y only contributes its parity
        cases.append((be(z), be(Q[0]), be(Q[1] ^ 1), be(rr) + be(ss)))        # wrong parity -> other point
        cases.append((be(z ^ 1), be(Q[0]), be(Q[1]), be(rr) + be(ss)))        # wrong digest
    Q = public_key(curve, 7)
    cases += [(be(5), be(Q[0]), be(Q[1]), be(0) + be(1)), (be(5), be(Q[0]), be(Q[1]), be(1) + be(c["n"])),   # r = 0, s = n
              (be(5), be(c["p"]), be(1), be(1) + be(1)), (be(5), be(5 if curve == 0 else 0), be(1), be(1) + be(1)),  # x >= p, x off curve
              (be(c["n"]), be(Q[0]), be(Q[1]), be(1) + be(1)), (be(1)[:31], be(Q[0]), be(Q[1]), be(1) + be(1))]  # z >= n, short digest
    for args in cases:
        assert ov(oracle, curve, *args) == verify(curve, *args), [a.hex() for a in args]
