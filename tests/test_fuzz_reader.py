"""Mutation fuzzing of the host code that parses untrusted bytes -- Circuit::read, the WitnessMap reader, and the planner that indexes by what
the bytes say -- under AddressSanitizer + UndefinedBehaviorSanitizer (`make asan`, CPU only). The reference's readers return Err on
malformed input, never UB (acir/src/circuit/mod.rs:154-161, native_types/witness_map.rs:108-146): every mutated input must end as
"parsed" or "refused"; a sanitizer report or a crash fails the test. Seeds: the reference's own byte-exact circuits
(tests/golden/reference_vectors.json) and circuits of every opcode kind from the generator; mutations act on the raw bincode (inside the
gzip layer, so that they reach the parser) and on the gzip bytes themselves."""
import gzip
import json
import os
import random
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "tools", "asan", "fuzz_driver")
N_MUTANTS = int(os.environ.get("ACVM_FUZZ_MUTANTS", "1500"))  # ~1 min under ASan; 30 000 ran clean in round 3 (ACVM_FUZZ_MUTANTS=30000)


@pytest.fixture(scope="module")
def driver():
    r = subprocess.run(["make", "-C", ROOT, "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return DRIVER


def seeds():
    from acvm_amd import synth
    from acvm_amd.acir import Brillig, Circuit, Expression as E, PermutationSort
    out = []
    with open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json")) as f:
        g = json.load(f)
    for v in g.get("serialization", {}).values() if isinstance(g.get("serialization"), dict) else []:
        if isinstance(v, dict) and "bytes" in v:
            out.append(bytes(v["bytes"]))
    for k in range(3):
        circ, _ = synth.mixed_circuit(120 + 40 * k, seed=0xF0220 + k)
        out.append(circ.to_bytes())
    circ, _ = synth.hash_circuit()
    out.append(circ.to_bytes())
    circ, _ = synth.grumpkin_circuit()
    out.append(circ.to_bytes())
    W = E.from_witness
    fc = Brillig(inputs=[W(1), [W(2), W(3)]], outputs=[4, [5, 6]], bytecode=[("ForeignCall", "f", [("Register", 0), ("HeapArray", 1, 2)], [("Register", 0), ("HeapVector", 1, 2)]),
                                                                               ("Call", 3), ("Stop",), ("Return",)])
    out.append(Circuit(9, [fc, PermutationSort([[W(1), W(2)], [W(3), W(4)]], 2, [7, 8], [0])]).to_bytes())
    return out


def mutate(r, raw):
    b = bytearray(raw)
    for _ in range(r.choice([1, 1, 1, 2, 3, 8])):
        if not b:
            break
        k = r.randrange(7)
        i = r.randrange(len(b))
        if k == 0:
            b[i] ^= 1 << r.randrange(8)
        elif k == 1:
            b[i] = r.choice([0, 1, 0x7F, 0x80, 0xFF, r.randrange(256)])
        elif k == 2 and len(b) >= 8:  # a length / index field: plant an extreme little-endian integer
            i = r.randrange(len(b) - 7)
            b[i:i + 8] = struct.pack("<Q", r.choice([0, 1, 2, 0xFFFFFFFF, 0x7FFFFFFF, 1 << 27, (1 << 27) - 1, 1 << 32, (1 << 64) - 1, r.randrange(1 << 20)]))
        elif k == 3 and len(b) >= 4:
            i = r.randrange(len(b) - 3)
            b[i:i + 4] = struct.pack("<I", r.choice([0, 13, 14, 0xFFFFFFFF, 0xFFFFFFFE, 1 << 27, r.randrange(64)]))
        elif k == 4:
            del b[i:i + r.choice([1, 4, 8, 64])]
        elif k == 5:
            b[i:i] = bytes(r.randrange(256) for _ in range(r.choice([1, 4, 8, 32])))
        else:
            b = b[:i]  # truncation
    return bytes(b)


def test_mutated_circuits_never_trip_the_sanitizers(driver, tmp_path):
    r = random.Random(0xACF022)
    blobs = []
    ss = seeds()
    assert len(ss) >= 6
    for s in ss:
        blobs.append(s)
        try:
            blobs.append(gzip.decompress(s))  # the reader takes raw bincode too
        except OSError:
            pass
    raws = [gzip.decompress(s) if s[:2] == b"\x1f\x8b" else s for s in ss]
    while len(blobs) < N_MUTANTS:
        raw = r.choice(raws)
        m = mutate(r, raw)
        which = r.randrange(10)
        if which < 6:
            blobs.append(m)                                   # raw bincode
        elif which < 9:
            blobs.append(gzip.compress(m, 1))                 # valid gzip around a mutated body
        else:
            blobs.append(mutate(r, gzip.compress(raw, 1)))    # a damaged gzip stream
    path = tmp_path / "blobs.bin"
    with open(path, "wb") as f:
        for b in blobs:
            f.write(struct.pack("<I", len(b)))
            f.write(b)
    env = dict(os.environ, ASAN_OPTIONS="abort_on_error=0:detect_leaks=1:allocator_may_return_null=1:max_allocation_size_mb=4096", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([driver, str(path)], capture_output=True, text=True, env=env, timeout=max(900, N_MUTANTS // 10))
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-6000:])
    assert "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr, out.stderr[-6000:]
    words = out.stdout.split()
    stats = dict(zip(words[0::2], map(int, words[1::2])))
    assert stats["blobs"] == len(blobs)
    # the seeds themselves parse and plan; a healthy share of the mutants still reaches the planner (the fuzzing is not all rejected at byte 0)
    assert stats["parsed"] >= 2 * len(ss) and stats["planned"] + stats["refused"] >= len(blobs) // 20, stats
    # every plan that built was laid out, scheduled and hazard-checked at two tile sizes under the sanitizers, without a finding
    assert stats["schedules"] == 2 * (stats["planned"] - stats["too_large_to_check"]) and stats["schedules"] >= stats["planned"] and stats["hazards"] == 0, (stats, out.stderr[-4000:])
