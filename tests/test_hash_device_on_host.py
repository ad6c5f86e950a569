"""The device hash routines (acvm_amd/csrc/hash_device.hpp is __host__ __device__: SHA-256, BLAKE2s-256 and Keccak-256 as the level and exact
kernels call them, blackbox_solver/src/lib.rs:47-65,86-99) executed on the host: SHA-256 and BLAKE2s against hashlib, Keccak-256 against the
oracle's (which the standard known answers pin, tests/test_oracle_hashes.py), over every message length around the block boundaries
(55/56/63/64/65 for SHA-256, 64/65 for BLAKE2s, 135/136/137 for Keccak's rate), the empty message and the reference's "hello world" vector
(brillig_vm/src/black_box.rs:203-208). No GPU is needed: hipcc builds the host side of tools/hash_device_host_test.hip, nothing is launched."""
import ctypes as C
import hashlib
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_device_hash_routines_on_host(tmp_path, oracle):
    exe = str(tmp_path / "hash_host_test")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O2", os.path.join(ROOT, "tools", "hash_device_host_test.hip"), "-o", exe], check=True, timeout=600)
    lengths = sorted(set(list(range(0, 70)) + [118, 119, 120, 127, 128, 129, 135, 136, 137, 138, 200, 271, 272, 273, 1023, 1024, 1025]))
    msgs = [bytes((7 * i + 13 * n + 1) & 0xFF for i in range(n)) for n in lengths] + [b"hello world", b"\xff" * 64, b"\x00" * 136]
    lines, want = [], []
    for m in msgs:
        for func in ("sha256", "blake2s", "keccak256"):
            lines.append(f"{func} {m.hex() or '-'}")
            if func == "sha256":
                want.append(hashlib.sha256(m).hexdigest())
            elif func == "blake2s":
                want.append(hashlib.blake2s(m).hexdigest())
            else:
                out32 = C.create_string_buffer(32)
                oracle.lib().oracle_keccak256(m, len(m), out32)
                want.append(out32.raw.hex())
    out = subprocess.run([exe], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=600)
    got = out.stdout.split()
    assert out.returncode == 0 and len(got) == len(want), out.stderr[-1000:]
    bad = [(lines[i][:40], got[i], want[i]) for i in range(len(want)) if got[i] != want[i]]
    assert not bad, bad[:5]
    assert got[lines.index("sha256 " + b"hello world".hex())] == "b94d27b9934d3e08a52e52d7da7dabfac484efe37a5380ee9088f7ace2efcde9"
