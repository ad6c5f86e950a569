"""Pins oracle/fr.c (restatement of acir_field::FieldElement) against the reference's own vectors and Python big-ints.
Reference vectors: generic_ark.rs:423-438 (hex of -0..-3, and idempotence), foreign_call.ts (5^-1),
witness_compression.ts (-1)."""
import random

P = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001


def be(x):
    return int(x).to_bytes(32, "big")


def op(oracle, code, a, b=0, aux=0):
    import ctypes as C
    out = C.create_string_buffer(32)
    oracle.lib().oracle_fr_op(code, be(a), be(b), aux, out)
    return int.from_bytes(out.raw, "big")


def test_negation_hex_vectors(oracle):
    # generic_ark.rs:426-431
    assert op(oracle, 4, 0) == 0
    assert op(oracle, 4, 1) == 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000000
    assert op(oracle, 4, 2) == 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593EFFFFFFF
    assert op(oracle, 4, 3) == 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593EFFFFFFE


def test_inverse_of_five(oracle, golden):
    expect = int(golden["acvm_js"]["foreign_call"]["oracleResponse"][0], 16)
    assert op(oracle, 5, 5) == expect == pow(5, P - 2, P)
    assert op(oracle, 5, 0) == 0  # inverse(0) == 0 (generic_ark.rs:242-245)
    assert op(oracle, 3, 7, 0) == 0  # a / 0 == 0


def test_and_idempotent(oracle):
    # generic_ark.rs:411-420: x.and(x, max_num_bits) == x for 0..10000 (sampled)
    for x in list(range(0, 300)) + [9999, 2**64 + 5, P - 1]:
        assert op(oracle, 6, x, x, 254) == x


def test_arithmetic_against_python(oracle):
    rng = random.Random(1)
    edge = [0, 1, 2, P - 1, P - 2, 2**128 - 1, 2**128, 2**128 + 1, 2**253, 2**253 + 12345, (P - 1) // 2]
    vals = edge + [rng.randrange(P) for _ in range(200)]
    for i, a in enumerate(vals):
        b = vals[(7 * i + 3) % len(vals)]
        assert op(oracle, 0, a, b) == (a + b) % P
        assert op(oracle, 1, a, b) == (a - b) % P
        assert op(oracle, 2, a, b) == (a * b) % P
        assert op(oracle, 3, a, b) == (a * pow(b, P - 2, P)) % P
        assert op(oracle, 5, a) == pow(a, P - 2, P)
        assert oracle.lib().oracle_fr_num_bits(be(a)) == a.bit_length()


def test_from_be_bytes_reduce(oracle):
    import ctypes as C
    rng = random.Random(2)
    for n in [0, 1, 2, 31, 32, 33, 48, 64, 100]:
        data = bytes(rng.randrange(256) for _ in range(n))
        out = C.create_string_buffer(32)
        oracle.lib().oracle_fr_from_bytes_reduce(data, n, out)
        assert int.from_bytes(out.raw, "big") == int.from_bytes(data, "big") % P
    out = C.create_string_buffer(32)
    oracle.lib().oracle_fr_from_bytes_reduce(b"\xff" * 32, 32, out)
    assert int.from_bytes(out.raw, "big") == (2**256 - 1) % P


def test_and_xor_masking(oracle):
    # generic_ark.rs:328-355,446-473: operands masked to the low num_bits first
    rng = random.Random(3)
    for _ in range(200):
        a, b = rng.randrange(P), rng.randrange(P)
        n = rng.choice([1, 7, 8, 9, 31, 32, 33, 64, 127, 128, 200, 253])
        m = (1 << n) - 1
        assert op(oracle, 6, a, b, n) == ((a & m) & (b & m)) % P
        assert op(oracle, 7, a, b, n) == ((a & m) ^ (b & m)) % P
    a, b = P - 1, P - 2  # num_bits >= 254 keeps everything, result reduced mod p
    assert op(oracle, 7, a, b, 254) == ((a & (2**254 - 1)) ^ (b & (2**254 - 1))) % P
    assert op(oracle, 6, a, b, 256) == (a & b) % P


def test_fetch_nearest_bytes(oracle):
    import ctypes as C
    # generic_ark.rs:305-317: ceil(bits/8) low bytes, least significant first
    x = 0x0102030405060708090A
    out = C.create_string_buffer(32)
    for bits, n in [(0, 0), (1, 1), (8, 1), (9, 2), (32, 4), (254, 32), (256, 32)]:
        got = oracle.lib().oracle_fr_fetch_nearest_bytes(be(x), bits, out)
        assert got == n
        assert out.raw[:n] == x.to_bytes(32, "little")[:n]
    assert oracle.lib().oracle_fr_fetch_nearest_bytes(be(x), 257, out) == -1  # reference: slice panic
