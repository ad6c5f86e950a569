"""BASELINE config 5 at circuit size on ONE GPU: the 10^6-opcode mixed circuit, solved in tiles of `tile`
instances through one reused batch handle; per tile only the results, the return witnesses and the per-instance digest of the
witness map are kept. An audit sample of the first tile is re-solved by the CPU oracle and compared bit-exactly (results,
return witnesses, and the digest recomputed with hashlib over the oracle's full map).

    python tools/t_config5.py [opcodes=1000000] [tile=4096] [n_tiles=2] [audit=8] [mode=plain|fold|reuse]
mode fold: the digest is computed during the solve (ACVM_BATCH_FOLD_DIGEST); reuse: witness-slot liveness reuse on top of it.
"""
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
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
n_tiles = int(sys.argv[3]) if len(sys.argv) > 3 else 2
audit = int(sys.argv[4]) if len(sys.argv) > 4 else 8
mode = sys.argv[5] if len(sys.argv) > 5 else "plain"

t0 = time.time()
circ, ids = synth.mixed_circuit(G)
data = circ.to_bytes()
t1 = time.time()
gc = acvm_amd.Circuit(data)
ret = gc.witness_set("return_values")
batch = acvm_amd.Batch(gc, tile, ids, fold_digest=mode in ("fold", "reuse"), reuse_slots=mode == "reuse", keep=ret)
t2 = time.time()
st0 = gc.plan_stats(ids)
row = len(ids) * 32
out = {"opcodes": G, "tile": tile, "mode": mode, "witnesses_per_instance": st0["n_witnesses"], "levels": st0["n_levels"],
       "witness_table_GB": round(st0["n_witnesses"] * 32 * tile / 1e9, 1), "generate_s": round(t1 - t0, 1), "parse_plan_alloc_s": round(t2 - t1, 1),
       "plan_ms": round(st0["plan_ms"]), "scaled_witnesses": st0["n_scaled_witnesses"], "tiles": []}
first_vals = None
for k in range(n_tiles):
    values = synth.witness_batch(tile, seed=0xAC1D0005, first_instance=k * tile)
    batch.set_profiling(k == n_tiles - 1 and n_tiles > 1)  # per-kernel HIP events on the last tile only (they cost time)
    batch.set_initial_witness(values)
    w0 = time.time()
    n_bad = batch.solve()
    w1 = time.time()
    dig = batch.digest()
    w2 = time.time()
    res = batch.results()
    solved = [j for j in range(tile) if res[j].status == 0]
    rv = batch.extract(ret, solved[0], 1) if solved and ret else None
    st = batch.stats()
    out["tiles"].append({"not_solved": n_bad, "solve_device_ms": round(st["solve_device_ms"], 1), "solve_wall_ms": round((w1 - w0) * 1e3, 1),
                         "digest_wall_ms": round((w2 - w1) * 1e3, 1), "launches": st["n_kernel_launches"], "slow_instances": st["n_slow_instances"],
                         "witnesses_per_s": round(tile / (w2 - w0), 1)})
    if k == n_tiles - 1 and n_tiles > 1:
        out["tiles"][-1]["kernel_ms"] = {"arith": round(st["arith_kernel_ms"], 1), "inverse_batch": round(st["dyn_kernel_ms"], 1),
                                         "light/hash/grumpkin/brillig": [round(x, 1) for x in st["class_kernel_ms"]]}
    if k == 0:
        picks = sorted(set([0, 5] + [int(x) for x in np.linspace(8, tile - 1, max(audit - 2, 1))]))[:audit]
        sub = b"".join(values[j * row:(j + 1) * row] for j in picks)
        a0 = time.time()
        ores, oasg, ovals = oracle.solve_batch(oracle.Circuit(data), ids, sub, len(picks), n_threads=min(len(picks), shard.cpu_budget()[0]))
        a1 = time.time()
        ok = True
        for i, j in enumerate(picks):
            ok &= res[j].as_tuple() == ores[i].as_tuple()
            ok &= bytes(dig[j]) == oracle.witness_map_digest(oasg[i], ovals[i])
            if ores[i].status == 0 and ret:
                got = batch.extract(ret, j, 1)[0]
                ok &= all(bytes(got[n]) == bytes(ovals[i][w]) for n, w in enumerate(ret))
        if mode != "reuse" and os.environ.get("ACVM_T_B2S"):  # the byte-wise tree digest (acvm_batch_digest_blake2s): its cost, and the audit instances against hashlib
            b0 = time.time()
            b2s = batch.digest_blake2s()
            out["blake2s_tree_digest_ms"] = round((time.time() - b0) * 1e3, 1)
            for i, j in enumerate(picks):
                ok &= bytes(b2s[j]) == oracle.witness_map_blake2s(oasg[i], ovals[i])
        out["audit"] = {"instances": picks, "bit_exact": bool(ok), "oracle_s": round(a1 - a0, 1),
                        "oracle_witnesses_per_s": round(len(picks) / (a1 - a0), 2), "oracle_threads": min(len(picks), shard.cpu_budget()[0])}
batch.free()
print(json.dumps(out))
