"""per-step series of solve_device_ms for import + solve of config 4 / ECDSA at 2^16: is the slow mode random, periodic, sticky?  python tools/t_step_series.py grumpkin [pipelined]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import acvm_amd  # noqa: E402
from acvm_amd import synth, tiling  # noqa: E402
wl = sys.argv[1]
pipe = len(sys.argv) > 2
B = 1 << 16
if wl == "grumpkin":
    circ, ids = synth.grumpkin_circuit()
    base = synth.grumpkin_rows(1024, first_instance=0)
    arr = np.frombuffer(synth.values_from_rows(base), dtype=np.uint8).reshape(len(base), -1)
    values = arr[np.arange(B) % len(base)].tobytes()
else:
    circ, ids = synth.ecdsa_circuit()
    values = synth.ecdsa_batch(B)
sh = tiling.ResidentShard(acvm_amd.Circuit(circ.to_bytes()), ids, values, B, B)
out = []
for i in range(60):
    sh.load_tile(0)
    sh.solve_tile(0, pipelined=pipe)
    out.append(sh.batch.stats()["solve_device_ms"])
print(wl, "pipelined" if pipe else "plain", " ".join(f"{x:.2f}" for x in out))
