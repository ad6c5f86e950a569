import sys, time
sys.path.insert(0, '.')
import acvm_amd
from acvm_amd import synth
B = 1 << 16
circ, ids = synth.arithmetic_circuit(10000, seed=0xAC1D0002)
batch = acvm_amd.Batch(acvm_amd.Circuit(circ.to_bytes()), B, ids)
batch.set_initial_witness(synth.witness_batch(B, seed=0xAC1D0002))
for prof in (True, False, True, False):
    batch.set_profiling(prof)
    batch.reset(); batch.solve()
    acvm_amd.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        batch.reset(); batch.solve()
    acvm_amd.synchronize()
    dt = (time.perf_counter() - t0) / 5
    st = batch.stats()
    print("profiling", prof, "ms/step %.2f" % (dt * 1e3), "device_ms %.2f" % st["solve_device_ms"], "arith %.2f dyn %.2f" % (st["arith_kernel_ms"], st["dyn_kernel_ms"]))
