"""Does what precedes a solve change the solve? Config 4 (Pedersen + FixedBase + Schnorr, 2^16 instances; ALU-bound, one wave per SIMD) timed by its own
HIP events (solve_device_ms) after: nothing (reset + solve back to back), an import of the resident inputs (bench.py's step), a host pause of
0.2 / 1 / 5 ms. The part drops its shader clock within microseconds of idling (239 MHz at rest) and takes a while to come back.
    python tools/t_step_gap.py [workload: grumpkin | ecdsa | hash]"""
import sys
import time

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import acvm_amd  # noqa: E402
from acvm_amd import synth, tiling  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "grumpkin"
B = 1 << 16
if wl == "grumpkin":
    circ, ids = synth.grumpkin_circuit()
    base = synth.grumpkin_rows(1024, first_instance=0)
    arr = np.frombuffer(synth.values_from_rows(base), dtype=np.uint8).reshape(len(base), -1)
    values = arr[np.arange(B) % len(base)].tobytes()
elif wl == "ecdsa":
    circ, ids = synth.ecdsa_circuit()
    values = synth.ecdsa_batch(B)
else:
    circ, ids = synth.hash_circuit()
    values = synth.byte_batch(B, len(ids))
sh = tiling.ResidentShard(acvm_amd.Circuit(circ.to_bytes()), ids, values, B, B)
b = sh.batch
sh.load_tile(0)
for _ in range(3):
    b.reset()
    b.solve()


def run(name, before, n=12):
    ms, wall = [], []
    for _ in range(n):
        before()
        t0 = time.perf_counter()
        b.solve()
        wall.append((time.perf_counter() - t0) * 1e3)
        ms.append(b.stats()["solve_device_ms"])
    ms.sort()
    wall.sort()
    print(f"{wl}: {name:42s} solve_device_ms median {ms[len(ms) // 2]:.3f} min {ms[0]:.3f} max {ms[-1]:.3f} | wall median {wall[len(wall) // 2]:.3f}", flush=True)


run("reset + solve, back to back", lambda: b.reset())
run("import (load_tile) + solve", lambda: sh.load_tile(0))
for pause in (0.0002, 0.001, 0.005, 0.02):
    run(f"reset, host pause {pause * 1e3:.1f} ms, solve", lambda: (b.reset(), time.sleep(pause)))
run("reset + solve, back to back (again)", lambda: b.reset())
# cache state: a streaming kernel over buffers the solve never touches (acvm_debug_stream_rate: three buffers of the given size, five passes)
for mb in (16, 64, 256, 1024):
    run(f"reset, {3 * mb} MB streamed through the caches, solve", lambda: (b.reset(), acvm_amd.stream_rate(mb << 20)), n=6)
run("reset + solve, back to back (again)", lambda: b.reset())
# what is it about the import? (a) an ALU kernel between the import and the solve; (b) the second solve behind one import
run("import, ALU probe kernel (modmul_rate 30 x 8), solve", lambda: (sh.load_tile(0), acvm_amd.modmul_rate(30, 8)), n=6)
run("reset, ALU probe kernel, solve", lambda: (b.reset(), acvm_amd.modmul_rate(30, 8)), n=6)
run("import, solve, reset, solve (the second one)", lambda: (sh.load_tile(0), b.solve(), b.reset()), n=6)
run("import x 3, solve", lambda: (sh.load_tile(0), sh.load_tile(0), sh.load_tile(0)), n=6)
run("import + solve", lambda: sh.load_tile(0), n=6)
# clocks? a LONG integer-bound run (4 x 0.73 ms of the product probe) between the import and the solve
run("import, 3 ms of the ALU probe, solve", lambda: (sh.load_tile(0), acvm_amd.modmul_rate(400, 8)), n=6)
run("import, 12 ms of the ALU probe, solve", lambda: (sh.load_tile(0), acvm_amd.modmul_rate(1600, 8)), n=4)
run("reset, 3 ms of the ALU probe, solve", lambda: (b.reset(), acvm_amd.modmul_rate(400, 8)), n=6)
# freshness of THIS table's rows, or what the import does to the caches? the same import into ANOTHER handle's table in front of the solve
sh2 = tiling.ResidentShard(acvm_amd.Circuit(circ.to_bytes()), ids, values, B, B)
sh2.load_tile(0)
run("reset, import into another handle's table, solve", lambda: (b.reset(), sh2.load_tile(0)), n=8)
run("import (this handle), solve", lambda: sh.load_tile(0), n=8)
run("reset + solve", lambda: b.reset(), n=8)
# a first READ of the fresh rows by a streaming kernel (RANGE(254) on every input: light records in front of the heavy launch) -- does the heavy kernel then find them "warm"?
from acvm_amd.acir import BlackBoxFuncCall as BB, Circuit, FunctionInput as FI  # noqa: E402
circ2 = Circuit(current_witness_index=circ.current_witness_index, opcodes=[BB("RANGE", {"input": FI(w, 254)}) for w in ids] + list(circ.opcodes),
                private_parameters=list(circ.private_parameters), return_values=list(circ.return_values))
sh3 = tiling.ResidentShard(acvm_amd.Circuit(circ2.to_bytes()), ids, values, B, B)
b3 = sh3.batch
sh3.load_tile(0)
b3.set_profiling(True)
for name, before in (("with RANGE on every input: reset + solve", lambda: b3.reset()), ("with RANGE on every input: import + solve", lambda: sh3.load_tile(0))):
    best = None
    for _ in range(8):
        before()
        b3.solve()
        st = b3.stats()
        if best is None or st["solve_device_ms"] < best["solve_device_ms"]:
            best = st
    print(f"{wl}: {name:48s} solve_device_ms {best['solve_device_ms']:.3f} class kernels ms {[round(x, 3) for x in best['class_kernel_ms']]}", flush=True)
