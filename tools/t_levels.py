"""Per-level overhead of the level loop: a chain circuit (one gate per level) at full batch size.
    python tools/t_levels.py [gates]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import acvm_amd
from acvm_amd import synth

gates = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
B = 1 << 16
for mix in ((47, 32, 21, 0), (45, 30, 20, 5)):
    circ, ids = synth.arithmetic_circuit(gates, seed=0xAC1D0077, chain=True, mix=mix)
    values = synth.witness_batch(B, seed=0xAC1D0077)
    gc = acvm_amd.Circuit(circ.to_bytes())
    batch = acvm_amd.Batch(gc, B, ids)
    best = None
    for it in range(4):
        batch.set_initial_witness(values)
        batch.solve()
        st = batch.stats()
        if best is None or st["solve_device_ms"] < best[0]:
            best = (st["solve_device_ms"], st["n_levels"], st["n_dyn_gates"], st["n_slow_instances"])
    print(f"chain mix={mix}: device {best[0]:.2f} ms, levels {best[1]}, dyn gates {best[2]}, slow instances {best[3]} -> {best[0] * 1e3 / best[1]:.1f} us per level")
    batch.free()
