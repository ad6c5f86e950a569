import time, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import acvm_amd
from acvm_amd import synth
circ, ids = synth.hash_circuit()
B = 1 << 16
vals = synth.byte_batch(B, len(ids))
b = acvm_amd.Batch(acvm_amd.Circuit(circ.to_bytes()), B, ids)
db = acvm_amd.DeviceBuffer(vals)
for i in range(3):
    b.set_initial_witness_device(db.ptr); b.solve()
ts = []
for i in range(20):
    b.set_initial_witness_device(db.ptr); acvm_amd.synchronize() if hasattr(acvm_amd, 'synchronize') else None
    t0 = time.perf_counter(); b.solve(); ts.append((time.perf_counter() - t0) * 1e3)
print("wall ms", sorted(ts)[:3], "median", sorted(ts)[10], "device ms", b.stats()["solve_device_ms"], {k: v for k, v in b.stats().items() if "launch" in k})
