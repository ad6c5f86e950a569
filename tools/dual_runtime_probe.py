"""Does torch's bundled HIP runtime coexist with libacvm_amd.so's system ROCm runtime in one process?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
print("torch", torch.__version__, "cuda available", torch.cuda.is_available())
x = torch.ones(1024, device="cuda") * 2
torch.cuda.synchronize()
print("torch sum", float(x.sum()))
import acvm_amd
from acvm_amd import synth
acvm_amd.set_device(0)
circ, ids = synth.arithmetic_circuit(200, seed=3)
b = acvm_amd.Batch(acvm_amd.Circuit(circ.to_bytes()), 128, ids)
b.set_initial_witness(synth.witness_batch(128, seed=3))
print("not solved:", b.solve(), b.stats()["solve_device_ms"])
y = torch.ones(1024, device="cuda") * 3
torch.cuda.synchronize()
print("torch again", float(y.sum()))
print("DUAL_RUNTIME_OK")
