"""GPU parity of EcdsaSecp256k1 / EcdsaSecp256r1 (acvm/src/pwg/blackbox/signature/ecdsa.rs, blackbox_solver/src/lib.rs:66-210):
the reference's two vectors (lib.rs:216-284) on the device, and seeded batches of valid / tampered / high-S / panicking inputs
against the CPU oracle, through the level kernel and the exact kernel."""
import random

import pytest

from acvm_amd.acir import BlackBoxFuncCall as BB, Circuit, FunctionInput as FI
from ecdsa_ref import CURVES, public_key, sign
from test_gpu_opcodes import both_paths
from test_oracle_ecdsa import K1, R1

pytestmark = pytest.mark.gpu


def ecdsa_circuit(curve, n_x=32, n_y=32, n_sig=64, n_msg=32):
    ids = list(range(1, n_x + n_y + n_sig + n_msg + 1))
    x, y, sig, msg = ids[:n_x], ids[n_x:n_x + n_y], ids[n_x + n_y:n_x + n_y + n_sig], ids[n_x + n_y + n_sig:]
    op = BB("EcdsaSecp256k1" if curve == 0 else "EcdsaSecp256r1",
            {"public_key_x": [FI(w, 8) for w in x], "public_key_y": [FI(w, 8) for w in y], "signature": [FI(w, 8) for w in sig],
             "hashed_message": [FI(w, 8) for w in msg], "output": ids[-1] + 1})
    return Circuit(ids[-1] + 1, [op]), ids


def row(x, y, sig, z):
    return list(x) + list(y) + list(sig) + list(z)


@pytest.mark.parametrize("curve,v", [(0, K1), (1, R1)])
def test_reference_vectors_on_device(oracle, curve, v):
    circ, ids = ecdsa_circuit(curve)
    good = row(*[bytes.fromhex(v[k]) for k in ("x", "y", "sig", "z")])
    bad = list(good)
    bad[70] ^= 1
    ores, _ = both_paths(oracle, circ, ids, [good, bad])
    assert ores[0].status == 0 and ores[1].status == 0
    import acvm_amd
    from acvm_amd.synth import values_from_rows
    batch = acvm_amd.Batch(acvm_amd.Circuit(circ.to_bytes()), 2, ids)
    batch.set_initial_witness(values_from_rows([good, bad]))
    assert batch.solve() == 0
    vals, asg = batch.witness(ids[-1] + 1)
    assert [int(vals[j, 31]) for j in range(2)] == [1, 0]


@pytest.mark.parametrize("curve", [0, 1])
def test_batch_against_oracle(oracle, curve):
    r = random.Random(40 + curve)
    c = CURVES[curve]
    be = lambda v: int(v).to_bytes(32, "big")  # noqa: E731
    rows = []
    for i in range(10):
        sk, k, z = r.randrange(1, c["n"]), r.randrange(1, c["n"]), r.randrange(c["n"])
        Q = public_key(curve, sk)
        rr, ss = sign(curve, sk, k, z, low_s=(i % 5 != 4))
        rows.append(row(be(Q[0]), be(Q[1]), be(rr) + be(ss), be(z)))
        rows.append(row(be(Q[0]), be(Q[1] ^ 2), be(rr) + be(ss), be(z)))       # only the parity of y matters
        rows.append(row(be(Q[0]), be(Q[1] ^ 1), be(rr) + be(ss), be(z)))
        rows.append(row(be(Q[0]), be(Q[1]), be(rr) + be(ss), be(z ^ 4)))
    Q = public_key(curve, 7)
    rows += [row(be(Q[0]), be(Q[1]), be(0) + be(1), be(5)), row(be(Q[0]), be(Q[1]), be(1) + be(c["n"]), be(5)),
             row(be(c["p"]), be(1), be(1) + be(1), be(5)), row(be(5 if curve == 0 else 0), be(1), be(1) + be(1), be(5)),
             row(be(Q[0]), be(Q[1]), be(1) + be(1), be(c["n"]))]
    rows.append([300 + b for b in rows[0]])  # witnesses above 255: only the last byte of each counts (to_u8_vec)
    circ, ids = ecdsa_circuit(curve)
    # the panic paths (rows 40..44): status, error kind and opcode are pinned by the call sites of blackbox_solver/src/lib.rs:120-129;
    # the message TEXT both sides report is UNPINNED (the reference's is k256 / p256's unwrap text)
    ores, _ = both_paths(oracle, circ, ids, rows)
    assert ores[0].status == 0 and ores[40].err == oracle.E_PANIC and ores[44].err == oracle.E_PANIC


@pytest.mark.parametrize("curve", [0, 1])
def test_batch_size_across_the_workgroup_boundary(oracle, curve):
    """the record kernel runs in workgroups of four waves (256 instances): 321 instances = one full workgroup, one full wave and a partly filled one"""
    from acvm_amd import synth
    import numpy as np
    v = (K1, R1)[curve]
    good = row(*[bytes.fromhex(v[k]) for k in ("x", "y", "sig", "z")])
    rows = []
    for i in range(321):
        r = list(good)
        if i % 5 == 3: r[128 + (i % 32)] ^= 1 << (i % 8)   # a flipped digest bit: invalid
        if i % 11 == 7: r[32 + 31] ^= 2                     # another y of the same parity: still the same key
        rows.append(r)
    circ, ids = ecdsa_circuit(curve)
    ores, _ = both_paths(oracle, circ, ids, rows)
    assert ores[0].status == 0


def test_wrong_lengths_and_short_digest(oracle):
    for kw in (dict(n_x=31), dict(n_y=33), dict(n_sig=63), dict(n_msg=31)):
        circ, ids = ecdsa_circuit(0, **kw)
        rows = [[1] * len(ids)] * 2
        ores, _ = both_paths(oracle, circ, ids, rows)
        assert ores[0].status == 2
