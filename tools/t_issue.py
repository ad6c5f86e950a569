"""Separate the ALU (issue) time of the arithmetic level kernels from their HBM time: the same gate mix with every operand
drawn from the 16 circuit inputs (reads hit L2, one dependency level), timed through the C ABI's own statistics.
    python tools/t_issue.py [gates] [mix as a,b,c,d]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import acvm_amd
from acvm_amd import synth

gates = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
mix = tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 else (45, 30, 20, 5)
B = 1 << 16
for hot in (True, False):
    circ, ids = synth.arithmetic_circuit(gates, seed=0xAC1D0002, mix=mix, hot_inputs=hot)
    values = synth.witness_batch(B, seed=0xAC1D0002)
    gc = acvm_amd.Circuit(circ.to_bytes())
    batch = acvm_amd.Batch(gc, B, ids)
    best = None
    for it in range(4):
        batch.set_initial_witness(values)
        t = time.perf_counter()
        batch.solve()
        dt = (time.perf_counter() - t) * 1e3
        st = batch.stats()
        if best is None or st["solve_device_ms"] < best[0]:
            best = (st["solve_device_ms"], st["arith_kernel_ms"], st["dyn_kernel_ms"], st["n_levels"], st["n_dyn_gates"], dt)
    print(f"hot_inputs={hot} mix={mix}: device {best[0]:.2f} ms, arith kernels {best[1]:.2f} ms, inversion kernels {best[2]:.2f} ms, "
          f"levels {best[3]}, dyn gates {best[4]}, wall {best[5]:.2f} ms")
    batch.free()
