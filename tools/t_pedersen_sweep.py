"""Pedersen only, the level kernel's two shapes (tuning pedersen_waves = 4 / 1) over launch sizes: N independent Pedersen{[a, b], 0} opcodes on one
level x B instances; ms per solve, best of 6.   python tools/t_pedersen_sweep.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import acvm_amd
from acvm_amd import synth
from acvm_amd.acir import BlackBoxFuncCall as BB, Circuit, FunctionInput as FI

ids = list(range(1, 17))
for B in (1 << 12, 1 << 14, 1 << 15, 1 << 16):
    values = synth.witness_batch(B, seed=0xAC1D0004, edge_cases=False)
    for n_rec in (1, 2, 4, 8):
        ops = [BB("Pedersen", {"inputs": [FI(1 + (2 * k) % 16, 254), FI(1 + (2 * k + 1) % 16, 254)], "domain_separator": 0, "outputs": [17 + 2 * k, 18 + 2 * k]}) for k in range(n_rec)]
        circ = Circuit(current_witness_index=16 + 2 * n_rec, opcodes=ops, private_parameters=ids)
        gc = acvm_amd.Circuit(circ.to_bytes())
        row = []
        for waves in (4, 1):
            with acvm_amd.tuning(pedersen_waves=waves):
                batch = acvm_amd.Batch(gc, B, ids)
                batch.set_initial_witness(values)
                best = 1e9
                for it in range(6):
                    batch.reset()
                    batch.solve()
                    best = min(best, batch.stats()["solve_device_ms"])
                batch.free()
            row.append(best)
        print(f"B {B:6d} records {n_rec}: groups {B // 64 * n_rec:6d}  four waves {row[0]:.3f} ms  one wave {row[1]:.3f} ms  ratio {row[1] / row[0]:.2f}", flush=True)
