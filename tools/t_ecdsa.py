"""EcdsaSecp256k1 / EcdsaSecp256r1 alone at batch 2^16: the reference's two vectors (blackbox_solver/src/lib.rs:216-284) replicated over
the batch, every 7th instance with a flipped message bit.   python tools/t_ecdsa.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import acvm_amd
from acvm_amd.acir import BlackBoxFuncCall as BB, Circuit, FunctionInput as FI
from test_oracle_ecdsa import K1, R1

B = 1 << 16
for curve, v in ((0, K1), (1, R1)):
    ids = list(range(1, 161))
    x, y, sig, msg = ids[:32], ids[32:64], ids[64:128], ids[128:]
    op = BB("EcdsaSecp256k1" if curve == 0 else "EcdsaSecp256r1",
            {"public_key_x": [FI(w, 8) for w in x], "public_key_y": [FI(w, 8) for w in y], "signature": [FI(w, 8) for w in sig],
             "hashed_message": [FI(w, 8) for w in msg], "output": 161})
    circ = Circuit(161, [op])
    good = np.frombuffer(b"".join(bytes.fromhex(v[k]) for k in ("x", "y", "sig", "z")), dtype=np.uint8)
    vals = np.zeros((B, 160, 32), dtype=np.uint8)
    vals[:, :, 31] = good[None, :]
    vals[::7, 140, 31] ^= 1
    batch = acvm_amd.Batch(acvm_amd.Circuit(circ.to_bytes()), B, ids)
    batch.set_initial_witness(vals.tobytes())
    best = 1e9
    for it in range(5):
        batch.reset()
        batch.solve()
        best = min(best, batch.stats()["solve_device_ms"])
    out, asg = batch.witness(161)
    ok = int(out[:, 31].sum())
    print(f"{'secp256k1' if curve == 0 else 'secp256r1'}: {best:.3f} ms per 65536 verifications, {ok} valid of {B} (expected {B - (B + 6) // 7}), {batch.stats()['n_slow_instances']} slow instances")
    batch.free()
