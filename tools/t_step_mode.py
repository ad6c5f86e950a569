"""one mode of tools/t_step_gap.py for a kernel trace: python tools/t_step_mode.py grumpkin|ecdsa reset|import"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import acvm_amd  # noqa: E402
from acvm_amd import synth, tiling  # noqa: E402
wl, mode = sys.argv[1], sys.argv[2]
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
b = sh.batch
sh.load_tile(0)
for _ in range(10):
    if mode == "import":
        sh.load_tile(0)
    else:
        b.reset()
    b.solve()
print(wl, mode, b.stats()["solve_device_ms"])
