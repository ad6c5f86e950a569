"""Config 3 (sha256 -> keccak256 of the digest + 32 more bytes, RANGE(8) on every input byte) at 2^16 / 2^17 / 2^18 instances and with K independent
copies of the chain on one level (K x 1 024 items per launch at 2^16): microseconds per launch of the byte-message hash kernel (HIP events of the
batch's profiling mode, best of the timed solves), the HBM fraction in algorithmic bytes, and the solve's device time -- the measurement DESIGN.md
section 9 asserted without ("the phases overlap once a launch holds more items than the device has slots").

    python tools/t_hash_sweep.py [tag]                       # prints one line per shape
    ACVM_TUNING=hash_chain=0 python tools/t_hash_sweep.py    # the unchained launches
"""
import sys
import time

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import acvm_amd  # noqa: E402
from acvm_amd import synth  # noqa: E402
from acvm_amd.acir import BlackBoxFuncCall as BB, Circuit, FunctionInput as FI  # noqa: E402


def chains_circuit(k_chains, n_msg=64):
    """k independent copies of config 3's chain over disjoint inputs: their SHA-256 records share a level, their Keccak records the next"""
    per = n_msg + 32
    ids = list(range(1, k_chains * per + 1))
    ops = [BB("RANGE", {"input": FI(w, 8)}) for w in ids]
    nxt = ids[-1] + 1
    ret = []
    for c in range(k_chains):
        mine = ids[c * per:(c + 1) * per]
        sha_out = list(range(nxt, nxt + 32))
        kec_out = list(range(nxt + 32, nxt + 64))
        nxt += 64
        ops.append(BB("SHA256", {"inputs": [FI(w, 8) for w in mine[:n_msg]], "outputs": sha_out}))
        ops.append(BB("Keccak256", {"inputs": [FI(w, 8) for w in sha_out + mine[n_msg:]], "outputs": kec_out}))
        ret += kec_out
    return Circuit(current_witness_index=nxt - 1, opcodes=ops, private_parameters=ids, return_values=ret), ids


def measure(k_chains, log2_b, reps=6):
    circ, ids = chains_circuit(k_chains)
    B = 1 << log2_b
    batch = acvm_amd.Batch(acvm_amd.Circuit(circ.to_bytes()), B, ids)
    batch.set_initial_witness(synth.byte_batch(B, len(ids)))
    batch.set_profiling(True)
    best = None
    for r in range(reps):
        batch.reset()
        failed = batch.solve()
        st = batch.stats()
        if r >= 2 and (best is None or st["class_kernel_ms"][1] < best["class_kernel_ms"][1]):
            best = st
    assert failed == 0
    batch.set_profiling(False)
    acvm_amd.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        batch.reset()
        batch.solve()
    acvm_amd.synchronize()
    wall = (time.perf_counter() - t0) / 5
    alg = best["class_algorithmic_bytes_per_instance"][1] * B
    ms = best["class_kernel_ms"][1]
    n_launch = best["n_kernel_launches"]
    out = (f"chains {k_chains} instances 2^{log2_b}: hash class {ms * 1e3:8.1f} us in {n_launch} launch(es), {alg / 1e6:8.1f} MB algorithmic = "
           f"{alg / (ms / 1e3) / 8e12:.3f} of 8 TB/s | solve device {best['solve_device_ms'] * 1e3:8.1f} us, wall {wall * 1e6:8.1f} us per solve | "
           f"items per launch {k_chains * B // 64}")
    batch.free()
    return out


if __name__ == "__main__":
    acvm_amd.set_device(0)
    print("tuning hash_chain =", acvm_amd.tuning_get("hash_chain") if hasattr(acvm_amd, "tuning_get") else "?")
    for k, lb in ((1, 16), (1, 17), (1, 18), (2, 16), (4, 16), (2, 17), (1, 15), (1, 14)):
        print(measure(k, lb), flush=True)
