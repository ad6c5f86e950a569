"""Deterministic synthetic ACIR circuits and witness batches for tests and the bench.

PRNG = splitmix64, seed 0xAC1D0000 + config number. Witness 0 is unused (Noir convention); inputs are
witnesses 1..n_in (all assigned); gate i (0-based) solves witness n_in+1+i. Operands are drawn uniformly
from already-defined witnesses. Gate mix (width-3 PLONK-shaped, acvm/src/compiler/transformers/csat.rs):
  45 %  qM*a*b + qo*out + qc
  30 %  q1*a + q2*b + qo*out + qc
  20 %  qM*a*b + q1*c + qo*out + qc
   5 %  qM*a*out + q1*c + qc        (unknown inside the mul term: per-instance inversion; zero multiplicand
                                     leaves the generic path, arithmetic.rs:217-221)
Coefficients: 50 % from {1, -1}, 50 % uniform Fr, never 0. Terms are sorted like Expression::sort
(acir/src/native_types/expression/mod.rs:176-179).
"""
import numpy as np

from .acir import BlackBoxFuncCall, Circuit, Expression, FunctionInput, P

MASK = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & MASK

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & MASK
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK
        return z ^ (z >> 31)

    def below(self, n):
        return self.next() % n

    def fr(self):
        v = 0
        for i in range(4):
            v |= self.next() << (64 * i)
        return v % P

    def coef(self):
        if self.next() & 1:
            return 1 if self.next() & 1 else P - 1
        while True:
            c = self.fr()
            if c:
                return c


def arithmetic_circuit(n_gates, n_in=16, seed=0xAC1D0002, chain=False, mix=(45, 30, 20, 5), hot_inputs=False):
    """Returns (Circuit, input witness ids). hot_inputs: operands are drawn from the circuit inputs only (one level, operand
    reads served by L2: the issue-bound variant used by tools/ to separate ALU time from HBM time)."""
    rng = SplitMix64(seed)
    ops = []
    defined = n_in  # witnesses 1..defined are known
    for i in range(n_gates):
        out = n_in + 1 + i
        pick = lambda: 1 + rng.below(n_in if hot_inputs else defined)  # noqa: E731
        a, b, c = pick(), pick(), pick()
        if chain and i > 0:
            a = out - 1
        r = rng.below(100)
        qc = rng.coef() if rng.next() & 1 else 0
        if r < mix[0]:
            e = Expression([(rng.coef(), a, b)], [(rng.coef(), out)], qc)
        elif r < mix[0] + mix[1]:
            if a == b:
                b = 1 + (b % defined)
            e = Expression([], [(rng.coef(), a), (rng.coef(), b), (rng.coef(), out)], qc)
        elif r < mix[0] + mix[1] + mix[2]:
            e = Expression([(rng.coef(), a, b)], [(rng.coef(), c), (rng.coef(), out)], qc)
        else:
            e = Expression([(rng.coef(), a, out)], [(rng.coef(), c)], qc)
        e.mul_terms.sort(key=lambda t: (t[1], t[2]))
        e.linear_combinations.sort(key=lambda t: t[1])
        ops.append(e)
        defined += 1
    circ = Circuit(current_witness_index=n_in + n_gates, opcodes=ops, private_parameters=list(range(1, n_in + 1)),
                   return_values=[n_in + n_gates])
    return circ, list(range(1, n_in + 1))


def wide_gate_circuit(n_gates, n_in=16, seed=0xAC1D0A11, max_terms=12):
    """Gates with up to max_terms mul terms and max_terms linear terms (Expression is not width-limited before the csat
    transformer), coefficients from {1, -1, uniform}: more than six general coefficients per gate, unit-coefficient
    terms beyond the device's side-sum budget, and every fourth gate followed by an assert-zero re-statement of it
    (all witnesses known: arithmetic.rs:92-102). Returns (Circuit, input ids)."""
    rng = SplitMix64(seed)
    ops = []
    defined = n_in
    out = n_in
    for i in range(n_gates):
        out += 1
        pick = lambda: 1 + rng.below(defined)  # noqa: E731
        nm, nl = rng.below(max_terms + 1), rng.below(max_terms + 1)
        unit_only = rng.below(3) == 0
        coef = (lambda: (1 if rng.next() & 1 else P - 1)) if unit_only else rng.coef
        mul = {}
        for _ in range(nm):
            a, b = pick(), pick()
            mul[(min(a, b), max(a, b))] = coef()
        lin = {}
        for _ in range(nl):
            lin[pick()] = coef()
        lin[out] = coef()
        qc = rng.coef() if rng.next() & 1 else 0
        e = Expression(sorted((c, a, b) for (a, b), c in mul.items()), sorted((c, w) for w, c in lin.items()), qc)
        e.mul_terms.sort(key=lambda t: (t[1], t[2]))
        e.linear_combinations.sort(key=lambda t: t[1])
        ops.append(e)
        defined = out
        if i % 4 == 3:
            ops.append(Expression(list(e.mul_terms), list(e.linear_combinations), qc))
    circ = Circuit(current_witness_index=out, opcodes=ops, private_parameters=list(range(1, n_in + 1)), return_values=[out])
    return circ, list(range(1, n_in + 1))


def _splitmix_vec(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def witness_batch(B, n_in=16, seed=0xAC1D0002, edge_cases=True, first_instance=0):
    """[B][n_in][32] big-endian input values as bytes. Values are raw 256-bit strings; both the reference
    (from_be_bytes_reduce) and the device import reduce them mod p. Instances 0..7 of the global batch are edge
    cases when edge_cases is set."""
    with np.errstate(over="ignore"):
        j = (np.arange(B, dtype=np.uint64) + np.uint64(first_instance))[:, None]
        k = np.arange(n_in, dtype=np.uint64)[None, :]
        base = np.uint64(seed) ^ ((j << np.uint64(16)) | k)
        limbs = np.empty((B, n_in, 4), dtype=np.uint64)
        s = base * np.uint64(4)
        for i in range(4):
            limbs[:, :, i] = _splitmix_vec(s + np.uint64(i))
    out = limbs.byteswap().view(np.uint8).reshape(B, n_in, 32).copy()  # each u64 big-endian, limb 0 most significant
    if edge_cases:
        pm1 = np.frombuffer((P - 1).to_bytes(32, "big"), dtype=np.uint8)
        one = np.frombuffer((1).to_bytes(32, "big"), dtype=np.uint8)
        two128 = np.frombuffer((1 << 128).to_bytes(32, "big"), dtype=np.uint8)
        for g in range(8):
            idx = g - first_instance
            if not (0 <= idx < B):
                continue
            if g == 0:
                out[idx] = 0
            elif g == 1:
                out[idx] = one
            elif g == 2:
                out[idx] = pm1
            elif g == 3:
                out[idx, 0::2] = 0
                out[idx, 1::2] = pm1
            elif g == 4:
                out[idx] = 0xFF  # 2^256 - 1: reduced on import
            elif g == 5:
                out[idx] = two128
            elif g == 6:
                out[idx, 0] = 0  # a single zero input
            elif g == 7:
                out[idx, :] = out[idx, 0]  # all inputs equal
    return out.tobytes()


# ------------------------------------------------------------------------------------------------ other configs
def be32(x: int) -> bytes:
    return int(x % (1 << 256)).to_bytes(32, "big")


def values_from_rows(rows) -> bytes:
    """rows: per instance, list of ints (one per initial witness, in id order) -> [B][n][32] big-endian bytes."""
    return b"".join(be32(v) for row in rows for v in row)


def hash_circuit(n_msg=64, with_range=True):
    """BASELINE config 3: n_msg byte-witnesses -> SHA256 -> 32 outputs; those 32 bytes + 32 more inputs ->
    Keccak256 (64-byte message) -> 32 outputs; RANGE(8) on every input byte. Returns (Circuit, input ids)."""
    from .acir import BlackBoxFuncCall as BB, FunctionInput as FI
    n_in = n_msg + 32
    ids = list(range(1, n_in + 1))
    ops = []
    if with_range:
        ops += [BB("RANGE", {"input": FI(w, 8)}) for w in ids]
    sha_out = list(range(n_in + 1, n_in + 33))
    kec_out = list(range(n_in + 33, n_in + 65))
    ops.append(BB("SHA256", {"inputs": [FI(w, 8) for w in ids[:n_msg]], "outputs": sha_out}))
    ops.append(BB("Keccak256", {"inputs": [FI(w, 8) for w in sha_out + ids[n_msg:]], "outputs": kec_out}))
    circ = Circuit(current_witness_index=kec_out[-1], opcodes=ops, private_parameters=ids, return_values=kec_out)
    return circ, ids


def byte_batch(B, n_in, seed=0xAC1D0003, first_instance=0):
    """[B][n_in][32]: uniform bytes as field elements."""
    with np.errstate(over="ignore"):
        j = (np.arange(B, dtype=np.uint64) + np.uint64(first_instance))[:, None]
        k = np.arange(n_in, dtype=np.uint64)[None, :]
        v = _splitmix_vec(np.uint64(seed) ^ ((j << np.uint64(20)) | k))
    out = np.zeros((B, n_in, 32), dtype=np.uint8)
    out[:, :, 31] = (v & np.uint64(0xFF)).astype(np.uint8)
    return out.tobytes()


Q_GRUMPKIN = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47


def grumpkin_circuit(msg_len=10, n_pedersen_inputs=2):
    """BASELINE config 4: Pedersen{[w1, w2], 0} -> FixedBaseScalarMul{low: w3, high: w4} -> SchnorrVerify
    (pk = w5, w6; 64 signature bytes w7..w70; msg_len message bytes). Returns (Circuit, input ids)."""
    from .acir import BlackBoxFuncCall as BB, FunctionInput as FI
    n_in = n_pedersen_inputs + 2 + 2 + 64 + msg_len
    ids = list(range(1, n_in + 1))
    o = n_pedersen_inputs
    ped_in, lo, hi, pkx, pky = ids[:o], ids[o], ids[o + 1], ids[o + 2], ids[o + 3]
    sig, msg = ids[o + 4:o + 68], ids[o + 68:]
    out = n_in + 1
    ops = [BB("Pedersen", {"inputs": [FI(w, 254) for w in ped_in], "domain_separator": 0, "outputs": [out, out + 1]}),
           BB("FixedBaseScalarMul", {"low": FI(lo, 128), "high": FI(hi, 128), "outputs": [out + 2, out + 3]}),
           BB("SchnorrVerify", {"public_key_x": FI(pkx, 254), "public_key_y": FI(pky, 254), "signature": [FI(w, 8) for w in sig],
                                "message": [FI(w, 8) for w in msg], "output": out + 4})]
    circ = Circuit(current_witness_index=out + 4, opcodes=ops, private_parameters=ids, return_values=[out, out + 1, out + 2, out + 3, out + 4])
    return circ, ids


def load_signed_fixture():
    """(pk || sig, msg) tuples of tests/golden/schnorr_signed.json (data made by the generator script next to it)."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "schnorr_signed.json")
    with open(path) as f:
        return [(bytes.fromhex(v["pk_sig"]), bytes.fromhex(v["msg"])) for v in json.load(f)["vectors"]]


def grumpkin_rows(B, signed=None, n_pedersen_inputs=2, seed=0xAC1D0004, first_instance=0):
    """Input rows for grumpkin_circuit. `signed`: list of (pk || sig (128 bytes), message) tuples, default the committed
    fixture; they are reused round-robin and odd instances get one flipped signature bit. Instances 0..7 of the global
    batch violate the limb / modulus checks."""
    signed = signed or load_signed_fixture()
    rows = []
    for t in range(B):
        j = first_instance + t
        r = SplitMix64(seed ^ (j * 0x9E3779B97F4A7C15 & MASK))
        ped = [r.fr() for _ in range(n_pedersen_inputs)]
        lo = r.next() | (r.next() << 64)
        hi = (r.next() | (r.next() << 64)) >> 3  # < 2^125: the scalar is always below the group order
        if j < 8:
            lo, hi = [(1 << 128, 0), (0, 1 << 128), (Q_GRUMPKIN & ((1 << 128) - 1), Q_GRUMPKIN >> 128), (P - 1, 1), (0, 0), (1, 0),
                      ((Q_GRUMPKIN - 1) & ((1 << 128) - 1), (Q_GRUMPKIN - 1) >> 128), (5, (1 << 128) - 1)][j]
        pk_sig, msg = signed[j % len(signed)]
        sig = bytearray(pk_sig[64:128])
        if j & 1:
            sig[r.below(64)] ^= 1 << r.below(8)
        rows.append(ped + [lo, hi, int.from_bytes(pk_sig[:32], "big"), int.from_bytes(pk_sig[32:64], "big")] + list(sig) + list(msg))
    return rows


def rows_to_bytes_fast(rows) -> bytes:
    return values_from_rows(rows)


# the two ECDSA vectors of the reference's own tests (blackbox_solver/src/lib.rs:216-284): hashed message, public key, signature
ECDSA_VECTORS = {
    "EcdsaSecp256k1": dict(z="3a73f4123a5cd2121f21cd7e8d358835476949d035d9c2da6806b4633ac8c1e2", x="a0434d9e47f3c86235477c7b1ae6ae5d3442d49b1943c2b752a68e2a47e247c7",
                           y="893aba425419bc27a3b6c7e693a24c696f794c2ed877a1593cbee53b037368d7",
                           sig="e5081c80ab427dc370346f4a0e31aa2bad8d9798c38061db9ae55a4e8df454fd28119894344e71b78770cc931d61f480ecbb0b89d6eb69690161e49a715fcd55"),
    "EcdsaSecp256r1": dict(z="54705ba3baafdbdfba8c5f9a70f7a89bee98d906b53e31074da7baecdc0da9ad", x="550f471003f3df97c3df506ac797f6721fb1a1fb7b8f6f83d224498a65c88e24",
                           y="136093d7012e509a73715cbd0b00a3cc0ff4b5c01b3ffa196ab1fb327036b8e6",
                           sig="2c70a8d084b62bfc5ce03641caf9f72ad4da8c81bfe6ec9487bb5e1bef62a13218ad9ee29eaf351fdc50f1520c425e9b908a07278b43b0ec7b872778c14e0784"),
}


def ecdsa_circuit():
    """one EcdsaSecp256k1 and one EcdsaSecp256r1 opcode (blackbox/signature/ecdsa.rs): 2 x 160 byte-wide inputs, one output each"""
    ops, ids = [], []
    w = 1
    for name in ("EcdsaSecp256k1", "EcdsaSecp256r1"):
        x, y, sig, msg = [list(range(w + a, w + b)) for a, b in ((0, 32), (32, 64), (64, 128), (128, 160))]
        ids += list(range(w, w + 160))
        w += 160
        ops.append((name, x, y, sig, msg))
    out0 = w
    circ = Circuit(out0 + 1, [BlackBoxFuncCall(name, {"public_key_x": [FunctionInput(v, 8) for v in x], "public_key_y": [FunctionInput(v, 8) for v in y],
                                                    "signature": [FunctionInput(v, 8) for v in sig], "hashed_message": [FunctionInput(v, 8) for v in msg],
                                                    "output": out0 + k}) for k, (name, x, y, sig, msg) in enumerate(ops)],
                   private_parameters=ids, return_values=[out0, out0 + 1])
    return circ, ids


def ecdsa_batch(B, first_instance=0):
    """the reference's two vectors in every instance; every 7th instance (of the global batch) has one message bit flipped: its signatures do not verify"""
    import numpy as np
    good = np.frombuffer(b"".join(bytes.fromhex(ECDSA_VECTORS[n][k]) for n in ("EcdsaSecp256k1", "EcdsaSecp256r1") for k in ("x", "y", "sig", "z")), dtype=np.uint8)
    vals = np.zeros((B, 320, 32), dtype=np.uint8)
    vals[:, :, 31] = good[None, :]
    bad = (np.arange(first_instance, first_instance + B) % 7) == 0
    vals[bad, 140, 31] ^= 1
    vals[bad, 300, 31] ^= 1
    return vals.tobytes()


def arith_pedersen_circuit(n_gates=10000, n_pedersen=8, n_in=16, seed=0xAC1D0006):
    """The north-star circuit shape: a width-3 arithmetic circuit of n_gates gates with n_pedersen Pedersen commitments
    (2 inputs each, taken from solved witnesses spread over the circuit) whose outputs feed later gates.
    Returns (Circuit, input ids)."""
    from .acir import BlackBoxFuncCall as BB, FunctionInput as FI
    circ, ids = arithmetic_circuit(n_gates, n_in=n_in, seed=seed)
    ops = list(circ.opcodes)
    nw = circ.current_witness_index
    rng = SplitMix64(seed ^ 0x5EED)
    extra = []
    for k in range(n_pedersen):
        # inputs: two witnesses solved in the first (k+1)/(n_pedersen+1) part of the circuit
        hi = n_in + max(2, (k + 1) * n_gates // (n_pedersen + 1))
        a, b = 1 + rng.below(hi), 1 + rng.below(hi)
        ox, oy = nw + 1, nw + 2
        nw += 2
        pos = hi - n_in  # insert right after the gate that solves witness `hi`
        extra.append((pos, BB("Pedersen", {"inputs": [FI(a, 254), FI(b, 254)], "domain_separator": 0, "outputs": [ox, oy]})))
        # one more gate that consumes the commitment: w = ox * oy + a
        out = nw + 1
        nw += 1
        extra.append((pos, Expression([(1, ox, oy)], [(1, a), (P - 1, out)], 0)))
    for pos, op in sorted(extra, key=lambda t: -t[0]):
        ops.insert(pos, op)
    # keep the relative order of the (Pedersen, consumer) pairs: sorted() is stable and both share `pos`, inserted in reverse
    fixed = []
    i = 0
    while i < len(ops):
        if i + 1 < len(ops) and isinstance(ops[i], Expression) and isinstance(ops[i + 1], BB) and ops[i + 1].name == "Pedersen" \
                and any(t[1] in ops[i + 1].args["outputs"] or t[2] in ops[i + 1].args["outputs"] for t in ops[i].mul_terms):
            fixed += [ops[i + 1], ops[i]]
            i += 2
        else:
            fixed.append(ops[i])
            i += 1
    return Circuit(current_witness_index=nw, opcodes=fixed, private_parameters=ids, return_values=[nw]), ids


def mixed_circuit(n_gates=2000, n_in=16, seed=0xAC1D0005, heavy=True, blocks=16, cells=64):
    """BASELINE config 5 shape at a chosen size: 94 % arithmetic gates of the config-2 mix whose operands are drawn
    uniformly from ALL witnesses defined so far (outputs of every opcode kind included), 3 % RANGE / AND / XOR, 1 % ToLeRadix(256,
    4 limbs) / Quotient, 1 % MemoryOp on `blocks` blocks of `cells` cells with a per-instance DYNAMIC index (a witness below
    `cells`), 0.5 % stdlib-shaped Brillig (BinaryIntOp then Stop), 0.5 % hash / Pedersen black boxes. Every opcode's outputs are
    fresh witnesses.

    Value classes the generator tracks so that the generic instance stays on the level kernels: `small` (< 2^8: radix digits, hash
    outputs), `w32` (< 2^32: results of 32-bit logic, the only sources ToLeRadix(256, 4) accepts), `idx` (< cells: AND of a byte with
    the constant cells - 1), `wide` (inputs and arithmetic outputs that are zero only with negligible odds: the multiplicand of an
    unknown-in-mul gate comes from here, a zero multiplicand sends an instance to the exact path, arithmetic.rs:217-221).
    Returns (Circuit, input ids)."""
    from .acir import BlackBoxFuncCall as BB, Brillig, FunctionInput as FI, MemoryInit, MemoryOp, QuotientDirective, ToLeRadix
    assert cells & (cells - 1) == 0 and cells <= 256
    rng = SplitMix64(seed)
    ops = []
    nw = n_in  # witnesses 1..nw are defined
    small, w32, idx = [], [], []
    wide = list(range(1, n_in + 1))
    is_wide = set(wide)

    def pick():  # any defined witness
        return 1 + rng.below(nw)

    def pick_wide():
        return wide[rng.below(len(wide))]

    def fresh(k=1):
        nonlocal nw
        out = list(range(nw + 1, nw + 1 + k))
        nw += k
        return out

    def logic(kind, lhs, rhs, bits):
        out, = fresh()
        ops.append(BB(kind, {"lhs": FI(lhs, bits), "rhs": FI(rhs, bits), "output": out}))
        return out

    # the constant cells - 1 as a witness (a gate without operands), a first 32-bit value, its four bytes, a first index
    cmask, = fresh()
    ops.append(Expression([], [(1, cmask)], P - (cells - 1)))
    w32.append(logic("AND", 1, 2, 32))
    d = fresh(4)
    ops.append(ToLeRadix(Expression.from_witness(w32[0]), d, 256))
    small += d
    idx.append(logic("AND", small[0], cmask, 8))
    for blk in range(blocks):
        ops.append(MemoryInit(blk, [pick_wide() for _ in range(cells)]))
    for i in range(n_gates):
        r = rng.below(1000)
        if r < 940:
            a, b, c = pick(), pick(), pick()
            out, = fresh()
            qc = rng.coef() if rng.next() & 1 else 0
            t = rng.below(100)
            # `terms`: the operand sets of the known terms; the output is zero only by coincidence ("wide") if the constant is a
            # random field element or some term has only wide operands -- unit coefficients on small operands DO cancel (1 * 1 - 1)
            if t < 45:
                e = Expression([(rng.coef(), a, b)], [(rng.coef(), out)], qc)
                terms = [(a, b)]
            elif t < 75:
                if a == b:
                    b = 1 + (b % (nw - 1))
                e = Expression([], [(rng.coef(), a), (rng.coef(), b), (rng.coef(), out)], qc)
                terms = [(a,), (b,)]
            elif t < 95:
                e = Expression([(rng.coef(), a, b)], [(rng.coef(), c), (rng.coef(), out)], qc)
                terms = [(a, b), (c,)]
            else:
                a = pick_wide()  # the known multiplicand of the unknown
                e = Expression([(rng.coef(), a, out)], [(rng.coef(), c)], qc)
                terms = [(c,)]
            e.mul_terms.sort(key=lambda t: (min(t[1], t[2]), max(t[1], t[2])))
            e.linear_combinations.sort(key=lambda t: t[1])
            ops.append(e)
            if qc not in (0, 1, P - 1) or any(all(u in is_wide for u in term) for term in terms):
                wide.append(out)
                is_wide.add(out)
        elif r < 970:
            k = rng.below(4)
            if k == 0:
                ops.append(BB("RANGE", {"input": FI(small[rng.below(len(small))], 8)}))
            elif k == 1:  # a memory index: byte AND (cells - 1)
                idx.append(logic("AND", small[rng.below(len(small))], cmask, 8))
            else:
                bits = [8, 32, 64, 254][rng.below(4)]
                out = logic("AND" if k == 2 else "XOR", pick(), pick(), bits)
                if bits == 32:
                    w32.append(out)
                elif bits == 8:
                    small.append(out)
        elif r < 980:
            if rng.next() & 1:
                d = fresh(4)
                ops.append(ToLeRadix(Expression.from_witness(w32[rng.below(len(w32))]), d, 256))
                small += d
            else:
                q, rem = fresh(2)
                ops.append(QuotientDirective(Expression.from_witness(pick()), Expression.from_witness(small[rng.below(len(small))]), q, rem))
        elif r < 990:
            blk = rng.below(blocks)
            index = Expression.from_witness(idx[rng.below(len(idx))])
            if rng.next() & 1:
                out, = fresh()
                ops.append(MemoryOp(blk, Expression.constant(0), index, Expression.from_witness(out)))
            else:
                ops.append(MemoryOp(blk, Expression.constant(1), index, Expression.from_witness(pick())))
        elif r < 995:
            op = ["Add", "Sub", "Mul", "UnsignedDiv"][rng.below(4)]
            bits = [32, 64, 127][rng.below(3)]
            lhs = pick()
            rhs = small[rng.below(len(small))] if op == "UnsignedDiv" else pick()
            out, = fresh()
            # stdlib shape (blackbox_fallbacks/uint.rs): r0 op= r1 at a fixed width; division guarded against 0 by + 1
            bc = [("Const", 2, 1), ("BinaryIntOp", 1, "Add", bits, 1, 2), ("BinaryIntOp", 0, op, bits, 0, 1), ("Stop",)] if op in ("UnsignedDiv", "Sub") \
                else [("BinaryIntOp", 0, op, bits, 0, 1), ("Stop",)]
            if op == "Sub":  # never underflows: (a mod 2^bits) - ((b + 1) mod 2^bits) wraps inside the VM
                bc = [("BinaryIntOp", 0, "Sub", bits, 0, 1), ("Stop",)] if bits >= 254 else [("BinaryIntOp", 0, "Add", bits, 0, 1), ("Stop",)]
            ops.append(Brillig(inputs=[Expression.from_witness(lhs), Expression.from_witness(rhs)], outputs=[out], bytecode=bc))
            if bits == 32:
                w32.append(out)
        elif heavy:
            k = rng.below(3)
            if k == 0:
                a, b = pick(), pick()
                ox, oy = fresh(2)
                ops.append(BB("Pedersen", {"inputs": [FI(a, 254), FI(b, 254)], "domain_separator": 0, "outputs": [ox, oy]}))
            else:
                n = 8 + rng.below(40)
                ins = [FI(small[rng.below(len(small))], 8) for _ in range(n)]
                outs = fresh(32)
                ops.append(BB("SHA256" if k == 1 else "Keccak256", {"inputs": ins, "outputs": outs}))
                small += outs
    circ = Circuit(current_witness_index=nw, opcodes=ops, private_parameters=list(range(1, n_in + 1)), return_values=[nw])
    return circ, list(range(1, n_in + 1))
