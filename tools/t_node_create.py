#!/usr/bin/env python3
"""tools/t_node_create.py [opcodes=1000000] [tile=512] -- what acvm_node_new costs for the 10^6-opcode circuit with 1 lane and with 8 lanes (device 0
listed eight times: the shape of an 8-GPU node rehearsed on one device): wall clock, the planner's own time, plans built, resident set. One plan per node
since round 6 (until then 1 + lanes plans: 9 x 2.3-2.5 s and nine copies of the plan)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import acvm_amd  # noqa: E402
from acvm_amd import synth  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 512
circ, ids = synth.mixed_circuit(G)
data = circ.to_bytes()
for lanes in (1, 8, 1, 8):
    gc = acvm_amd.Circuit(data)  # a fresh circuit handle: nothing cached
    ret = gc.witness_set("return_values")
    t0 = time.time()
    node = acvm_amd.Node(gc, ids, keep=ret, devices=[0] * lanes, tile=tile, reuse_slots=True)
    wall = time.time() - t0
    st = node.stats()
    values = synth.witness_batch(tile * lanes, seed=0xAC1D0005, first_instance=4096)
    ns, _, kept, asg, dig = node.solve(values, tile * lanes, results=False)
    print(json.dumps({"lanes": lanes, "tile": tile, "create_wall_s": round(wall, 2), "create_ms": round(st["create_ms"]), "plan_ms": round(st["plan_ms"]), "plans_built": st["plans_built"],
                      "circuit_plans_built": gc.plans_built(), "host_rss_GB": round(st["host_rss_bytes"] / 1e9, 2), "not_solved": ns}), flush=True)
    node.free()
    del node, gc
