"""What a tile of the 10^6-opcode circuit (BASELINE config 5) costs when k of its instances leave the level path (the hand-over to the
exact path multiplies / gathers every scaled row of every flagged instance). The first instances of synth.witness_batch are its edge cases (zeros,
p - 1, ...), which this circuit's generic path cannot finish: their rows are copied over k instances spread through the tile.

    python tools/t_c5_failures.py [opcodes=1000000] [tile=4096] [mode=plain|reuse] [k list = 0,1,8,64,512]
Prints one JSON line: per k the device time of the solve, the part behind the level schedule (hand-over + exact kernels) and whether the flagged
instances' results equal those of the instances they were copied from.
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import acvm_amd
from acvm_amd import synth

G = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
mode = sys.argv[3] if len(sys.argv) > 3 else "plain"
ks = [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "0,1,8,64,512").split(",")]

circ, ids = synth.mixed_circuit(G)
gc = acvm_amd.Circuit(circ.to_bytes())
ret = gc.witness_set("return_values")
batch = acvm_amd.Batch(gc, tile, ids, fold_digest=mode == "reuse", reuse_slots=mode == "reuse", keep=ret)
row = len(ids) * 32
base = bytearray(synth.witness_batch(tile, seed=0xAC1D0005, first_instance=tile))  # a tile without edge cases
edge = synth.witness_batch(8, seed=0xAC1D0005, first_instance=0)
batch.set_initial_witness(bytes(edge) + bytes(base[8 * row:]))
batch.solve()
res = batch.results()
bad = [j for j in range(8) if res[j].status != 0]
want = {j: res[j].as_tuple() for j in bad}
out = {"opcodes": G, "tile": tile, "mode": mode, "edge_instances": bad, "scaled_witnesses": gc.plan_stats(ids)["n_scaled_witnesses"], "runs": []}
for k in ks:
    vals = bytearray(base)
    where = [int(x) for x in np.linspace(3, tile - 3, k)] if k else []
    for n, j in enumerate(where):
        src = bad[n % len(bad)]
        vals[j * row:(j + 1) * row] = edge[src * row:(src + 1) * row]
    ms = []
    for rep in range(2):  # (the first run of a larger k grows the side table)
        batch.set_initial_witness(bytes(vals))
        w0 = time.time()
        n_bad = batch.solve()
        w1 = time.time()
        res = batch.results()  # (waits for the exact lanes, which run on the side stream behind the hand-over)
        w2 = time.time()
        st = batch.stats()
        ms.append((round(st["solve_device_ms"], 1), round(st["slow_path_ms"], 1), round((w1 - w0) * 1e3, 1), round((w2 - w1) * 1e3, 1)))
    same = all(res[j].as_tuple()[:1] == want[bad[n % len(bad)]][:1] for n, j in enumerate(where))
    out["runs"].append({"k": k, "flagged": st["n_slow_instances"], "not_solved": n_bad, "solve_device_ms": [m[0] for m in ms], "hand_over_ms": [m[1] for m in ms],
                        "solve_wall_ms": [m[2] for m in ms], "results_wall_ms": [m[3] for m in ms],
                        "status_as_source": bool(same)})
    print(json.dumps(out["runs"][-1]), file=sys.stderr, flush=True)
print(json.dumps(out))
