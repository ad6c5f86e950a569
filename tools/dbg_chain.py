"""Debug helper: first witness that differs between the HIP path and the oracle on the chain circuit."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import acvm_amd
from acvm_amd import synth
from oracle import binding as oracle
oracle.build(); oracle.lib()
circ, ids = synth.arithmetic_circuit(300, seed=0xAC1D0077, chain=True)
B = 64
values = synth.witness_batch(B, seed=0xAC1D0077)
data = circ.to_bytes()
oc = oracle.Circuit(data)
ores, oasg, ovals = oracle.solve_batch(oc, ids, values, B)
gc = acvm_amd.Circuit(data)
batch = acvm_amd.Batch(gc, B, ids)
batch.set_initial_witness(values)
batch.solve()
gres = batch.results()
gasg, gvals = batch.witness_map()
for j in range(B):
    if gres[j].as_tuple() != ores[j].as_tuple():
        print("instance", j, "gpu", gres[j].as_tuple(), "oracle", ores[j].as_tuple())
nw = min(oasg.shape[1], gasg.shape[1])
for j in range(B):
    for w in range(nw):
        if oasg[j, w] != gasg[j, w] or (oasg[j, w] and not np.array_equal(ovals[j, w], gvals[j, w])):
            op = circ.opcodes[w - 17] if w >= 17 else None
            print("first diff instance", j, "witness", w, "asg", oasg[j, w], gasg[j, w])
            print(" oracle", bytes(ovals[j, w]).hex()); print(" gpu   ", bytes(gvals[j, w]).hex())
            if op is not None:
                print(" gate mul", [(hex(c)[:10], a, b) for c, a, b in op.mul_terms], "lin", [(hex(c)[:10], a) for c, a in op.linear_combinations], "qc", hex(op.q_c)[:10])
            break
    else:
        continue
    break
for j in (0, 3):
    for w in (96, 97):
        print("inst", j, "w", w, "oracle", oasg[j, w], bytes(ovals[j, w]).hex()[:24], "gpu", gasg[j, w], bytes(gvals[j, w]).hex()[:24])
for k in (79, 80):
    op = circ.opcodes[k]
    print(k, "gate mul", [(hex(c)[:10], a, b) for c, a, b in op.mul_terms], "lin", [(hex(c)[:10], a) for c, a in op.linear_combinations], "qc", hex(op.q_c)[:10])
st = batch.stats(); print({k: st[k] for k in st if 'level' in k or 'slow' in k or 'dyn' in k})
