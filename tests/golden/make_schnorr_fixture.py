#!/usr/bin/env python3
"""Generates tests/golden/schnorr_signed.json: 16 (secret key, nonce, message) -> (public key, signature) tuples made with
the CPU oracle's Schnorr signer (oracle/grumpkin.c oracle_schnorr_sign, the signer matching SURVEY Appendix A.3). The
bench and the tests use them as DATA for the SchnorrVerify inputs of BASELINE config 4; the verifier under test never sees
the signer. Run from the repo root:  python tests/golden/make_schnorr_fixture.py"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from acvm_amd.synth import Q_GRUMPKIN, SplitMix64, be32  # noqa: E402
from oracle import binding as ob  # noqa: E402


def main():
    rng = SplitMix64(0xAC1D0004)
    out = []
    for _ in range(16):
        sk = rng.fr() % Q_GRUMPKIN or 1
        k = rng.fr() % Q_GRUMPKIN or 1
        msg = bytes(rng.below(256) for _ in range(10))
        buf = C.create_string_buffer(128)
        assert ob.lib().oracle_schnorr_sign(be32(sk), be32(k), msg, len(msg), buf) == 0
        assert ob.lib().oracle_schnorr_verify(buf.raw[:64], buf.raw[64:128], 64, msg, len(msg)) == 1
        out.append({"sk": "%064x" % sk, "k": "%064x" % k, "msg": msg.hex(), "pk_sig": buf.raw.hex()})
    with open(os.path.join(ROOT, "tests", "golden", "schnorr_signed.json"), "w") as f:
        json.dump({"source": "tests/golden/make_schnorr_fixture.py (oracle_schnorr_sign)", "vectors": out}, f, indent=1)
    print("wrote", len(out), "vectors")


if __name__ == "__main__":
    main()
