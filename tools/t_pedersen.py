"""Pedersen only: N independent Pedersen{[a, b], 0} opcodes on one level, batch 2^16 (the shape of the level kernel inside the north-star
circuit), timed back to back; with `pmc` as argument it is meant to run under rocprofv3 --pmc (tools/gpu_pmc_cmd.sh).
    python tools/t_pedersen.py [n_records]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import acvm_amd
from acvm_amd import synth
from acvm_amd.acir import BlackBoxFuncCall as BB, Circuit, FunctionInput as FI

n_rec = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = 1 << 16
ids = list(range(1, 17))
ops = [BB("Pedersen", {"inputs": [FI(1 + (2 * k) % 16, 254), FI(1 + (2 * k + 1) % 16, 254)], "domain_separator": 0, "outputs": [17 + 2 * k, 18 + 2 * k]}) for k in range(n_rec)]
circ = Circuit(current_witness_index=16 + 2 * n_rec, opcodes=ops, private_parameters=ids)
values = synth.witness_batch(B, seed=0xAC1D0004, edge_cases=False)
batch = acvm_amd.Batch(acvm_amd.Circuit(circ.to_bytes()), B, ids)
batch.set_initial_witness(values)
best = 1e9
for it in range(6):
    batch.reset()
    batch.solve()
    best = min(best, batch.stats()["solve_device_ms"])
print(f"{n_rec} Pedersen records: {best:.3f} ms per 65536 instances = {best / n_rec:.3f} ms per record, {batch.stats()['n_slow_instances']} slow instances")
