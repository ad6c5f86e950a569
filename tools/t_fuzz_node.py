#!/usr/bin/env python3
"""tools/t_fuzz_node.py [n_seeds] -- seeded sweep of the node-level driver (acvm_node_solve) against one plain batch of the same instances:
config-5 style circuits with edge-case inputs, random batch sizes, tiles, handle counts (device 0 listed 1..3 times) and planner flags;
results, kept witnesses, assigned flags and digests must agree bit for bit, twice per handle (the second call re-uses the staging and the
side tables). The plain batch itself is what tools/t_fuzz.py holds against the oracle. Exits 1 on the first divergence."""
import os
import random
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import acvm_amd  # noqa: E402
from acvm_amd import synth  # noqa: E402
from test_gpu_node import plain_batch  # noqa: E402

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
t0 = time.time()
for seed in range(n_seeds):
    r = random.Random(0xD00D + seed)
    n_ops = r.choice([200, 400, 700, 1100])
    B = r.choice([1, 63, 64, 65, 130, 257, 600, 1000])
    tile = r.choice([64, 128, 192, 256, 512])
    handles = r.choice([1, 1, 2, 3])
    mode = r.choice(["plain", "plain", "fold", "reuse"])
    circ, ids = synth.mixed_circuit(n_ops, seed=0x5EED0000 + seed)
    values = synth.witness_batch(B, seed=0x5EED0000 + seed, edge_cases=r.random() < 0.7)
    data = circ.to_bytes()
    gc = acvm_amd.Circuit(data)
    keep = gc.witness_set("return_values") + [ids[0]]
    want = plain_batch(data, ids, values, B, keep)
    try:
        node = acvm_amd.Node(gc, ids, keep=keep, devices=[0] * handles, tile=tile, fold_digest=mode in ("fold", "reuse"), reuse_slots=mode == "reuse")
    except acvm_amd.AcvmError as e:
        print(f"seed {seed}: ops {n_ops} B {B} tile {tile} handles {handles} {mode}: refused ({str(e)[:50]})")
        continue
    for rep in range(2):
        not_solved, res, kept, asg, dig = node.solve(values, B)
        ok = ([x.as_tuple() for x in res] == want[0] and np.array_equal(asg, want[2]) and np.array_equal(kept, want[1]) and np.array_equal(dig, want[3])
              and not_solved == sum(1 for x in want[0] if x[0] != 0))
        if not ok:
            print(f"seed {seed}: ops {n_ops} B {B} tile {tile} handles {handles} {mode} rep {rep}: DIVERGES")
            sys.exit(1)
    st = node.stats()
    assert sum(st["exact_instances"]) <= B
    print(f"seed {seed}: ops {n_ops} B {B} tile {tile} handles {handles} {mode}: ok (not solved {not_solved}, exact {sum(st['exact_instances'])}, async {st['async_exact']})", flush=True)
    node.free()
print(f"{n_seeds} node configurations bit-exact against one batch in {time.time() - t0:.0f} s")
