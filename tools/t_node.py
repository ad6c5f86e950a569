"""tools/t_node.py [opcodes=1000000] [instances=16384] [tile=4096] [handles=1] [flags=plain|fold|reuse] [audit=4] [reps=3] -- the config-5 mixed circuit through the
node-level driver (acvm_node_*) on ONE GPU: `handles` batch handles on device 0, each driven by its own host thread, tiles of `tile`
instances; host-resident inputs, per instance the result, the return witness and the map digest come back. Prints witnesses/s of the
whole call (uploads, solves, exact path, exports) and the per-handle statistics; audits a sample against the CPU oracle."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import acvm_amd
from acvm_amd import shard, synth
from oracle import binding as oracle  # checker of the audit sample only

G = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
tile = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
handles = int(sys.argv[4]) if len(sys.argv) > 4 else 1
mode = sys.argv[5] if len(sys.argv) > 5 else "plain"
audit = int(sys.argv[6]) if len(sys.argv) > 6 else 4
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 3
circ, ids = synth.mixed_circuit(G)
data = circ.to_bytes()
gc = acvm_amd.Circuit(data)
ret = gc.witness_set("return_values")
t0 = time.time()
node = acvm_amd.Node(gc, ids, keep=ret, devices=[0] * handles, tile=tile, fold_digest=mode in ("fold", "reuse"), reuse_slots=mode == "reuse")
t1 = time.time()
values = synth.witness_batch(N, seed=0xAC1D0005)
out = {"opcodes": G, "instances": N, "tile": tile, "handles": handles, "mode": mode, "create_s": round(t1 - t0, 1), "runs": []}
for rep in range(reps):
    w0 = time.time()
    not_solved, res, kept, asg, dig = node.solve(values, N)
    w1 = time.time()
    st = node.stats()
    out["runs"].append({"wall_ms": round((w1 - w0) * 1e3, 1), "witnesses_per_s": round(N / (w1 - w0), 1), "not_solved": not_solved,
                        "solve_device_ms": [round(x, 1) for x in st["solve_device_ms"]], "h2d_wait_ms": [round(x, 1) for x in st["h2d_wait_ms"]],
                        "export_ms": [round(x, 1) for x in st["export_ms"]], "exact_instances": st["exact_instances"], "tiles": st["tiles"]})
# the audit sample (SURVEY 8d asks for 256 instances at the per-GPU share): the edge-case instances 0, 5, 8, the last one, the rest spread evenly
picks = sorted(set([0, 5, 8, N - 1] + [int(x) for x in np.linspace(9, N - 2, max(audit - 4, 0))]))[:max(audit, 4)]
row = len(ids) * 32
sub = b"".join(values[j * row:(j + 1) * row] for j in picks)
threads = min(len(picks), shard.cpu_budget()[0])  # (the cgroup quota, not the host's CPU count)
a0 = time.time()
ores, oasg, ovals = oracle.solve_batch(oracle.Circuit(data), ids, sub, len(picks), n_threads=threads)
a1 = time.time()
ok = all(res[j].as_tuple() == ores[i].as_tuple() and bytes(dig[j]) == oracle.witness_map_digest(oasg[i], ovals[i]) for i, j in enumerate(picks))
ok = ok and all(bytes(kept[j][n]) == bytes(ovals[i][w]) for i, j in enumerate(picks) if ores[i].status == 0 for n, w in enumerate(ret))
out["audit_bit_exact"] = bool(ok)
out["audit"] = {"instances": len(picks), "oracle_threads": threads, "oracle_s": round(a1 - a0, 1), "oracle_witnesses_per_s": round(len(picks) / (a1 - a0), 2),
                "not_solved_in_sample": sum(1 for r in ores if r.status != 0), "checked": "result records, kept (return) witnesses, map digests"}
st = node.stats()
out["placement"] = {"numa_node": st["numa_node"], "n_cpus_pinned": st["n_cpus_pinned"], "first_cpu": st["first_cpu"]}
print(json.dumps(out))
