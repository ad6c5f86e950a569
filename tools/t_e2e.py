"""tools/t_e2e.py [total_log2=20] -- the metric's workload (10k-gate arithmetic circuit, 2^20 instances from host memory) through acvm_node_solve with
one and with two batch handles on the device and tiles of 2^16 / 2^17: witnesses/s of the whole call, per-handle statistics."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import acvm_amd
import bench

total = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 20)
a = argparse.Namespace(workload="arith", gates=10000, pedersen=8)
circ, ids, values, wname = bench.make_workload(a, 0, total)
gc = acvm_amd.Circuit(circ.to_bytes())
ret = gc.witness_set("return_values")
for handles, tile in ((1, 1 << 17), (2, 1 << 17), (2, 1 << 16), (1, 1 << 16), (3, 1 << 16)):
    node = acvm_amd.Node(gc, ids, keep=ret, devices=[acvm_amd.current_device()] * handles, tile=tile)
    node.solve(values[: tile * handles * len(ids) * 32], tile * handles, results=False, digests=False)
    runs = []
    for rep in range(3):
        t0 = time.perf_counter()
        not_solved, _, kept, asg, dig = node.solve(values, total, results=False, digests=False)
        runs.append(time.perf_counter() - t0)
    st = node.stats()
    print(json.dumps({"handles": handles, "tile": tile, "witnesses_per_s": [round(total / r) for r in runs], "not_solved": not_solved,
                      "solve_device_ms": [round(x, 1) for x in st["solve_device_ms"]], "h2d_wait_ms": [round(x, 1) for x in st["h2d_wait_ms"]],
                      "export_ms": [round(x, 1) for x in st["export_ms"]]}), flush=True)
    node.free()
