"""Per-opcode cost of the Grumpkin black boxes at batch 2^16: each opcode of BASELINE config 4 alone in a circuit.
    python tools/t_grumpkin.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import acvm_amd
from acvm_amd import synth
from acvm_amd.acir import Circuit

B = 1 << 16
circ, ids = synth.grumpkin_circuit()
rows = synth.grumpkin_rows(B)
values = synth.values_from_rows(rows)
for k, name in enumerate(["Pedersen", "FixedBaseScalarMul", "SchnorrVerify"]):
    one = Circuit(current_witness_index=circ.current_witness_index, opcodes=[circ.opcodes[k]], private_parameters=ids)
    batch = acvm_amd.Batch(acvm_amd.Circuit(one.to_bytes()), B, ids)
    best = 1e9
    batch.set_initial_witness(values)
    for it in range(6):  # back to back, like bench.py (the first solves run at a lower clock)
        batch.reset()
        batch.solve()
        best = min(best, batch.stats()["solve_device_ms"])
    print(f"{name}: {best:.3f} ms per 65536 instances, {batch.stats()['n_slow_instances']} slow instances")
    batch.free()
