"""A 10k-gate arithmetic circuit with ONE oracle call per instance in the middle (a Brillig ForeignCall whose answer feeds the second
half): time of the two solves around the call when the answered batch re-enters the level schedule (default) and when every instance
continues on the exact in-order kernels (ACVM_FC_RELEVEL=0), and the host time of resolving.

    python tools/t_foreign.py [gates=10000] [B=8192]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import acvm_amd
from acvm_amd import synth
from acvm_amd.acir import P, Brillig, Circuit, Expression as E

G = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
circ, ids = synth.arithmetic_circuit(G, seed=0xAC1D0002)
ops = list(circ.opcodes)
nw = circ.current_witness_index
mid = G // 2
src = 16 + mid            # the witness gate mid - 1 solved
oracle_out = nw + 1       # fresh witness: the oracle's answer
ops.insert(mid, Brillig(inputs=[E.from_witness(src)], outputs=[oracle_out],
                        bytecode=[("ForeignCall", "square", [("Register", 0)], [("Register", 0)]), ("Stop",)]))
# one gate of the second half consumes the answer: w = answer * src + w_prev
consumer = nw + 2
ops.insert(mid + 1, E([(1, oracle_out, src)], [(1, src - 1), (P - 1, consumer)], 0))
circ = Circuit(current_witness_index=consumer, opcodes=ops, private_parameters=ids, return_values=[consumer])
data = circ.to_bytes()
values = synth.witness_batch(B, seed=0xAC1D0002, edge_cases=False)
out = {"gates": G, "instances": B}
for mode in ("relevel", "exact"):
    acvm_amd.tuning_set("fc_relevel", 1 if mode == "relevel" else 0)
    batch = acvm_amd.Batch(acvm_amd.Circuit(data), B, ids)
    batch.set_initial_witness(values)
    t0 = time.perf_counter()
    batch.solve()
    t1 = time.perf_counter()
    waiting = [j for j, r in enumerate(batch.results()) if r.status == acvm_amd.STATUS_REQUIRES_FOREIGN_CALL]
    t2 = time.perf_counter()
    for j in waiting:
        fn, inputs = batch.get_pending_foreign_call(j)
        batch.resolve_pending_foreign_call(j, [inputs[0][0] * inputs[0][0] % P])
    t3 = time.perf_counter()
    n_bad = batch.solve()
    t4 = time.perf_counter()
    v, a = batch.witness(consumer)
    out[mode] = {"solve1_ms": round((t1 - t0) * 1e3, 1), "waiting": len(waiting), "resolve_host_ms": round((t3 - t2) * 1e3, 1),
                 "solve2_ms": round((t4 - t3) * 1e3, 1), "not_solved": n_bad, "exact_lanes_after": batch.stats()["n_slow_instances"],
                 "consumer_sum": int(v[:, 24:].astype(np.uint64).sum())}
    batch.free()
assert out["relevel"]["consumer_sum"] == out["exact"]["consumer_sum"]
print(json.dumps(out))
