"""Where a step of the metric's workload goes: per tile of 2^17 instances, the solve's device time (HIP events), the exact path's share, and the wall
clock of load + solve -- with the next tile's import behind the solve (acvm_batch_solve_then_import) and without.   python tools/t_tile_gap.py"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import acvm_amd  # noqa: E402
from acvm_amd import synth, tiling  # noqa: E402

total, tile = 1 << 20, 1 << 17
circ, ids = synth.arithmetic_circuit(10000, seed=0xAC1D0002)
values = synth.witness_batch(total, seed=0xAC1D0002)
sh = tiling.ResidentShard(acvm_amd.Circuit(circ.to_bytes()), ids, values, total, tile)
for _ in range(2):
    sh.solve_pass()
for pipelined in (True, False, True, False):
    acvm_amd.synchronize()
    rows = []
    t_step = time.perf_counter()
    for k in range(len(sh.starts)):
        t0 = time.perf_counter()
        sh.load_tile(k)
        t1 = time.perf_counter()
        sh.solve_tile(k, pipelined=pipelined)
        t2 = time.perf_counter()
        st = sh.batch.stats()
        rows.append((t1 - t0, t2 - t1, st["solve_device_ms"], st["slow_path_ms"], st["n_slow_instances"]))
    acvm_amd.synchronize()
    step = (time.perf_counter() - t_step) * 1e3
    print(f"pipelined={pipelined}: step {step:.2f} ms; sum of solve_device_ms {sum(r[2] for r in rows):.2f}; exact path {sum(r[3] for r in rows):.2f} ms")
    print("   per tile: load ms " + " ".join(f"{r[0] * 1e3:.3f}" for r in rows))
    print("             solve wall ms " + " ".join(f"{r[1] * 1e3:.2f}" for r in rows))
    print("             solve device ms " + " ".join(f"{r[2]:.2f}" for r in rows) + " | flagged " + " ".join(str(r[4]) for r in rows))
