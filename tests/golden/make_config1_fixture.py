#!/usr/bin/env python3
"""Generates tests/golden/config1.json: BASELINE config 1 (1 000-gate arithmetic circuit, one instance) solved
with PLAIN PYTHON INTEGERS -- an in-order restatement of ArithmeticSolver::solve for this generator's gate shapes
(acvm/src/pwg/arithmetic.rs:27-127), independent of both the C oracle and the kernels -- and stored as data: the input
values, a few named witnesses, and the SHA-256 of all solved witness values in index order. tests/test_oracle_acvm.py
checks the oracle against it, tests/test_gpu_parity.py the HIP path. Run from the repo root:
    python tests/golden/make_config1_fixture.py"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from acvm_amd import synth  # noqa: E402
from acvm_amd.acir import P  # noqa: E402

SEED = 0xAC1D0001
INSTANCE = 9  # the first non-edge-case instance of the seeded batch


def solve(circ, w):
    """In-order solve; every gate of the generator has exactly one unknown."""
    for k, e in enumerate(circ.opcodes):
        const = e.q_c
        unknown, coef = None, 0
        for c, a, b in e.mul_terms:
            if a in w and b in w:
                const += c * w[a] * w[b]
            elif a in w or b in w:
                known, unk = (a, b) if a in w else (b, a)
                assert unknown in (None, unk)
                unknown, coef = unk, coef + c * w[known]
            else:
                raise AssertionError("two unknowns in a mul term")
        for c, a in e.linear_combinations:
            if a in w:
                const += c * w[a]
            else:
                assert unknown in (None, a)
                unknown, coef = a, coef + c
        coef %= P
        assert unknown is not None and coef != 0, f"gate {k} is not solvable for this instance"
        w[unknown] = (-const) * pow(coef, -1, P) % P
    return w


def main():
    circ, ids = synth.arithmetic_circuit(1000, seed=SEED)
    values = synth.witness_batch(INSTANCE + 1, seed=SEED)
    row = values[INSTANCE * len(ids) * 32:(INSTANCE + 1) * len(ids) * 32]
    w = {wid: int.from_bytes(row[32 * k:32 * k + 32], "big") % P for k, wid in enumerate(ids)}
    solve(circ, w)
    nw = circ.current_witness_index
    assert sorted(w) == list(range(1, nw + 1))
    digest = hashlib.sha256(b"".join(w[i].to_bytes(32, "big") for i in range(1, nw + 1))).hexdigest()
    out = {"source": "tests/golden/make_config1_fixture.py (Python big-integer solve)", "seed": SEED, "gates": 1000, "instance": INSTANCE,
           "inputs_be32_hex": row.hex(), "n_witnesses": nw, "sha256_of_witnesses_1_to_n": digest,
           "witnesses": {str(i): "%064x" % w[i] for i in (17, 18, 100, 516, 1000, nw)}}
    with open(os.path.join(ROOT, "tests", "golden", "config1.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote config1.json", digest)


if __name__ == "__main__":
    main()
