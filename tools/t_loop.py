"""Solve the config-2 circuit over and over for a few seconds (clock / power sampling beside it, tools/gpu_clock_cmd.sh):
    python tools/t_loop.py [seconds] [hot]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import acvm_amd
from acvm_amd import synth

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
hot = len(sys.argv) > 2 and sys.argv[2] == "hot"
B = 1 << 16
circ, ids = synth.arithmetic_circuit(10000, seed=0xAC1D0002, hot_inputs=hot)
values = synth.witness_batch(B, seed=0xAC1D0002)
batch = acvm_amd.Batch(acvm_amd.Circuit(circ.to_bytes()), B, ids)
t0 = time.perf_counter()
n, best = 0, 1e9
while time.perf_counter() - t0 < secs:
    batch.set_initial_witness(values)
    batch.solve()
    best = min(best, batch.stats()["solve_device_ms"])
    n += 1
print(f"{n} solves, best {best:.2f} ms per 2^16 instances")
