#!/usr/bin/env python3
"""tools/t_c5_sweep.py [opcodes=1000000] [n_tiles=4] -- the config-5 tile (10^6-opcode circuit, slot reuse, folded digest) under a list of
tuning settings and tile sizes in ONE process: the circuit is generated once, every setting gets its own plan and handle; per setting the
median solve_device_ms of the tiles after the first, witnesses/s, and the Blake2s of the tile's digests (the same for every setting: a
setting that changes a result shows at once). Settings: ACVM_SWEEP="tile=4096;tile=8192,pedersen_epoch=4;..." or the built-in list."""
import hashlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import acvm_amd  # noqa: E402
from acvm_amd import synth  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n_tiles = int(sys.argv[2]) if len(sys.argv) > 2 else 4
DEFAULT = ["tile=4096", "tile=8192", "tile=8192,pedersen_bundle_waves=1024", "tile=8192,pedersen_bundle_waves=512", "tile=8192,pedersen_epoch=4", "tile=8192,pedersen_epoch=8,pedersen_latency=4",
           "tile=8192,pedersen_epoch=8,pedersen_latency=4,pedersen_bundle_waves=512", "tile=8192,digest_epoch=16", "tile=8192,digest_epoch=32", "tile=8192,inv_epoch=8", "tile=8192,pedersen_bundle=2", "tile=8192,pedersen_epoch=4,pedersen_latency=3,digest_epoch=16"]
specs = os.environ.get("ACVM_SWEEP", "").split(";") if os.environ.get("ACVM_SWEEP") else DEFAULT
circ, ids = synth.mixed_circuit(G)
data = circ.to_bytes()
gc = acvm_amd.Circuit(data)
ret = gc.witness_set("return_values")
base_digest = {}
for spec in specs:
    kv = dict(item.split("=") for item in spec.split(",") if item)
    tile = int(kv.pop("tile", 4096))
    mode = {k: int(v) for k, v in kv.items()}
    with acvm_amd.tuning(**mode):
        t0 = time.time()
        batch = acvm_amd.Batch(gc, tile, ids, reuse_slots=True, keep=ret)
        create_s = time.time() - t0
        ms, wall = [], []
        dig0 = None
        for k in range(n_tiles):
            values = synth.witness_batch(tile, seed=0xAC1D0005, first_instance=4096 + k * tile)  # (no edge-case instances: they sit in [0, 8))
            batch.set_initial_witness(values)
            w0 = time.time()
            bad = batch.solve()
            dig = batch.digest()
            wall.append((time.time() - w0) * 1e3)
            ms.append(batch.stats()["solve_device_ms"])
            if k == 0:
                dig0 = hashlib.blake2s(bytes(dig[:4096].tobytes())).hexdigest()[:16]
        st = batch.stats()
        batch.free()
    base_digest.setdefault(4096, dig0)
    m = sorted(ms[1:])[len(ms[1:]) // 2]
    w = sorted(wall[1:])[len(wall[1:]) // 2]
    print(json.dumps({"spec": spec, "tile": tile, "solve_device_ms": round(m, 2), "solve+digest_wall_ms": round(w, 2), "witnesses_per_s": round(tile / (w / 1e3)), "per_4096_ms": round(w * 4096 / tile, 2),
                      "levels": st["n_levels"], "launches": st["n_kernel_launches"], "not_solved_last": bad, "create_s": round(create_s, 1), "digest_first_4096": dig0,
                      "same_results": dig0 == base_digest[4096]}), flush=True)
