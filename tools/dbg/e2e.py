import sys, time, os
sys.path.insert(0, "/root/repo")
import numpy as np
import acvm_amd
from acvm_amd import synth, tiling
circ, ids = synth.arithmetic_circuit(10000, seed=0xAC1D0002)
N = 1 << 20
tile = 1 << 17
values = synth.witness_batch(N, seed=0xAC1D0002)
gc = acvm_amd.Circuit(circ.to_bytes())
ret = gc.witness_set("return_values")
for label, kw, edge in (("node", {}, True),):
  for tune in ({}, {"exact_async": 0}):
   with acvm_amd.tuning(**tune):
    print(tune)
    node = acvm_amd.Node(gc, ids, keep=ret, devices=[0], tile=tile)
    node.solve(values[: tile * len(ids) * 32], tile, results=False)
    for rep in range(2):
        t0 = time.perf_counter()
        r = node.solve(values, N, results=False, digests=False)
        dt = time.perf_counter() - t0
        st = node.stats()
        print(label, rep, round(dt * 1e3, 1), "ms", {k: [round(x, 1) for x in st[k]] for k in ("solve_device_ms", "h2d_wait_ms", "export_ms", "lane_ms")}, st["exact_instances"], flush=True)
    node.free()
# the same through the resident-shard loop of bench.py
sh = tiling.ResidentShard(gc, ids, values, N, tile)
for rep in range(3):
    t0 = time.perf_counter(); dev = 0
    for k in range(len(sh.starts)):
        sh.load_tile(k); sh.batch.solve(); dev += sh.batch.stats()["solve_device_ms"]
    acvm_amd.synchronize()
    print("resident", rep, round((time.perf_counter() - t0) * 1e3, 1), "ms device", round(dev, 1), flush=True)
# resident loop + a concurrent H2D stream of the same size from a helper thread
import threading
buf = acvm_amd.DeviceBuffer(size=tile * len(ids) * 32)
host = np.frombuffer(values, dtype=np.uint8)
def up(k):
    acvm_amd.set_device(0)
    buf.upload(host[k * tile * 512:(k + 1) * tile * 512])
for rep in range(2):
    t0 = time.perf_counter(); dev = 0
    for k in range(len(sh.starts)):
        th = threading.Thread(target=up, args=(k,)); th.start()
        sh.load_tile(k); sh.batch.solve(); dev += sh.batch.stats()["solve_device_ms"]
        th.join()
    print("resident+h2d", rep, round((time.perf_counter() - t0) * 1e3, 1), "ms device", round(dev, 1), flush=True)
