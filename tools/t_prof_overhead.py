import sys, os
sys.path.insert(0, os.getcwd())
import acvm_amd
from acvm_amd import synth
circ, ids = synth.arithmetic_circuit(10000, seed=0xAC1D0002)
B = 1 << 16
values = synth.witness_batch(B, seed=0xAC1D0002)
gc = acvm_amd.Circuit(circ.to_bytes())
batch = acvm_amd.Batch(gc, B, ids)
batch.set_initial_witness(values)
for prof in (True, False, True, False):
    batch.set_profiling(prof)
    ts = []
    for it in range(5):
        batch.reset(); batch.solve(); ts.append(batch.stats()["solve_device_ms"])
    print("profiling", prof, ["%.2f" % t for t in ts])

# does a busy device before the first solve change the first solves? (clock ramp vs first touch)
import time
batch.free()
t = time.time(); bad = acvm_amd.selftest(1 << 22, 7); print("selftest 4M lanes: %.0f ms, mismatches %d" % ((time.time() - t) * 1e3, bad))
batch = acvm_amd.Batch(gc, B, ids)
batch.set_initial_witness(values)
batch.set_profiling(False)
ts = []
for it in range(6):
    batch.reset(); batch.solve(); ts.append(batch.stats()["solve_device_ms"])
print("fresh batch after a busy device, no profiling", ["%.2f" % t for t in ts])
